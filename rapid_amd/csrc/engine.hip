// librapid_mi355x.so -- host side of the C ABI declared in include/rapid_mi355x.h.
//
// One rapid_engine = one simulated cluster on one MI355X: the registry of endpoints, the K-ring view built
// on the device, the resident alert streams of the simulated receivers, and the per-round outputs.  All
// device work is enqueued on the engine's own HIP stream.  There is NO CPU fallback: every entry point
// that needs the device returns RAPID_EDEVICE when it is not a usable gfx950.
#include <hip/hip_runtime.h>
#include <pthread.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <set>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_segmented_radix_sort.hpp>
#include <rccl/rccl.h>

#if !defined(RAPID_TEST_BUILD) && (defined(RAPID_MEASUREMENT_BUILD) || defined(RAPID_PHASE_TIMERS) || defined(RAPID_BLOCK_STAMPS) || defined(RAPID_TRACE))
#error "measurement hooks (tally_probes.inc) are for test / measurement builds only: the product library is compiled without them"
#endif

#include "../../include/rapid_mi355x.h"
#include "index_kernels.h"
#include "node_id_sort.h"
#include "tally_kernel.h"
#include "view_kernels.h"
#include "vote_kernels.h"

namespace {

// Measurement knobs read from the environment exist in the TEST build only (-DRAPID_TEST_BUILD: librapid_mi355x_test.so, the
// same sources plus the rapid_debug_* entry points and the probe kernels); the product library reads no environment variable.
#ifdef RAPID_TEST_BUILD
inline const char* env_knob(const char* name) { return getenv(name); }
#else
inline const char* env_knob(const char*) { return nullptr; }
#endif

template <class T>
struct DevBuf {
    T* p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t n) {
        if (n <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        const size_t want = n + n / 8 + 64;
        hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), want * sizeof(T));
        if (e == hipSuccess) cap = want;
#ifdef RAPID_TEST_BUILD
        // RAPID_POISON=<byte>: every fresh device buffer is filled with that byte, so that a kernel reading scratch nobody wrote
        // misbehaves on every box, not only on one whose memory a previous tenant left dirty (the round-5 fault hunt)
        if (e == hipSuccess)
            if (const char* v = getenv("RAPID_POISON")) {
                e = hipMemset(p, (int)strtol(v, nullptr, 0) & 255, want * sizeof(T));
                if (e == hipSuccess) e = hipDeviceSynchronize();
            }
#endif
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

}  // namespace

struct rapid_engine {
    rapid_engine_config cfg{};
    int max_cut = 0;
    hipStream_t stream = nullptr;
    std::string err;

    // ---- registry (host) ----
    int n_nodes = 0;
    std::vector<int64_t> id_hi, id_lo;
    std::vector<uint8_t> reg_blob;  // the registry as given to rapid_view_build (+ rapid_view_register_endpoints): hostname bytes,
    std::vector<int> reg_off, reg_ports;  // offsets [n_nodes + 1], ports
    std::vector<uint8_t> member;  // host mirror
    int n_members = 0;
    // identifiersSeen (R/MembershipView.java:474-500: ordered by signed high, then signed low == NodeIdComparator; never
    // pruned) lives SORTED ON THE DEVICE (d_ids_hi / d_ids_lo, n_ids_dev entries): membership questions are binary searches
    // there, and the NodeIds a change admits wait in ids_pending until rebuild_view merges them in
    std::vector<std::pair<int64_t, int64_t>> ids_pending;
    bool view_built = false;
    int64_t config_id = -1;

    // ---- view (device) ----
    DevBuf<unsigned char> d_blob;
    DevBuf<int> d_host_off, d_ports;
    DevBuf<long long> d_keys;  // [K][n_nodes]
    DevBuf<unsigned long long> d_hx_host0, d_hx_port0;
    DevBuf<unsigned char> d_member;
    DevBuf<int> d_members;                       // member indices, ascending
    DevBuf<unsigned long long> d_sort_keys;      // [K][M] unsorted
    DevBuf<int> d_sort_vals;                     // [K][M]
    DevBuf<unsigned long long> d_ring_skeys;     // [K][M] sorted sortable keys
    DevBuf<int> d_ring;                          // [K][M] node indices in ring order
    DevBuf<int> d_pos;                           // [K][n_nodes]
    DevBuf<int> d_obs, d_subj;                   // [n_nodes][K]
    DevBuf<long long> d_ids_hi, d_ids_lo, d_ids_hi2, d_ids_lo2, d_ids_new;
    DevBuf<long long> d_cfg_out;
    DevBuf<unsigned long long> d_cfg_partial;
    DevBuf<int> d_chunk_kept, d_joiners, d_join_nodes, d_join_vals;   // incremental view change (view_kernels.h)
    DevBuf<unsigned long long> d_join_keys, d_join_skeys;
    DevBuf<unsigned long long> d_bitmaps;  // [receivers][bitmap_words]: the voters' proposals over the round's hot slots (launch_tally)
    int bitmap_words = 0;
    DevBuf<unsigned char> d_sort_tmp;
    DevBuf<int> d_seg_off;                       // [K + 1] ring boundaries inside the [K][M] sort buffers
    std::vector<int> seg_host;                   // its host copy (lives as long as the async upload needs it)
    std::vector<uint8_t> ring_member;            // member flags the device rings were built from (empty: no rings yet)
    // nodes whose member flag was set or cleared since then (a superset, repeats allowed), as long as changed_valid: a view change
    // then costs the host what the change is, not a walk over the whole registry (10^6 nodes: 1 ms of a 2.4 ms view change)
    std::vector<int> changed;
    bool changed_valid = false;
    DevBuf<int> d_chunk_base, d_chunk_lb, d_nonmembers;  // ring_chunk_prep_kernel's answers; nonmember_list_kernel's list (count first)
    DevBuf<unsigned int> d_idacc;                // ids_contains_publish_kernel: {answer, finished workgroups}
    bool idacc_clean = false;
    std::vector<unsigned int> change_stamp;      // per node: the view change that last listed it (a node named twice in one cut is listed once)
    unsigned int change_epoch = 0;
    DevBuf<int> d_gone;                          // the nodes that left, for the kernels of a view change
    DevBuf<unsigned short> d_edges, d_edge_mask; // index_edges_kernel: every hot slot's observer slots and ring mask (large populations)
    int ring_m = 0;                              // their length
    int n_ids_dev = 0;

    // Q4: the reference's memoised getObserversOf (R/MembershipView.java:210-224) on the device -- d_q4_rows[node][K] is what a
    // member's observers were when it was first hot since its entry was last dropped, d_q4_valid[node] says whether an entry
    // exists; the round index reads hot members' observers from here (index_kernels.h), rebuild_view drops what ringAdd /
    // ringDelete drop.  q4_emulate == false: the index always reads today's table.
    DevBuf<int> d_q4_nodes, d_q4_rows;
    DevBuf<unsigned char> d_q4_valid, d_q4_flag;
    bool q4_emulate = true;
    bool q4_live = false;  // the last index build found a hot member whose memoised observers are not today's

    // host mirrors of the tables (filled lazily after a rebuild)
    bool host_tables_valid = false;
    std::vector<int> h_obs, h_subj, h_ring;
    std::vector<long long> h_keys;
    // the ring-0 keys alone (what the ring-0 order of a decided cut needs, R/MembershipService.java:346-348): a function of the
    // registered endpoints, not of who is a member -- they stay valid over view changes, unlike the tables above
    std::vector<long long> h_keys0;
    bool host_keys0_valid = false;

    // ---- simulated population ----
    // The delivered streams = the records the tally reads, receiver after receiver (d_rec_off).  rec_fmt == kFmtBoundary: the
    // 20-byte rapid_alert_records themselves, exactly as they crossed the boundary -- copied into d_records_own by
    // rapid_sim_load_streams / _device, or borrowed in place (rapid_sim_attach_streams_device) -- read from HBM ONCE per round, by
    // the tally kernel.  rec_fmt == kFmtResident: the 8-byte resolved records rapid_sim_generate writes into d_records_own.
    DevBuf<unsigned char> d_records_own;
    const unsigned char* d_records = nullptr;
    int rec_fmt = rapid::kFmtBoundary;
    unsigned long long records_bytes = 0;  // readable bytes at d_records
    // generated streams carry entries of the round index of the view they were generated in (gen_cfg_id); gen_clean: every
    // alert of the set carried that view's configuration id and named a registered node
    long long gen_cfg_id = 0;
    bool gen_clean = false;
    bool offsets_on_device_only = false;  // attached: n_records_total is an upper bound until total_records() has fetched it
    DevBuf<long long> d_rec_off_own;
    const long long* d_rec_off = nullptr;
    int n_receivers = 0;
    bool streams_loaded = false, tallied = false;
    int force_exact = 0;
    DevBuf<int> d_emit, d_nprop, d_pcount, d_props;
    DevBuf<unsigned long long> d_fp, d_stats;
    DevBuf<unsigned int> d_next;

    // ---- per-round index over the loaded streams (index_kernels.h) ----
    bool index_valid = false;
    long long n_records_total = 0;
    DevBuf<unsigned char> d_alert_set;  // the round's distinct alerts, if the host declared them (uploaded; d_alerts points at it)
    const unsigned char* d_alerts = nullptr;  // ... or at the caller's device buffer (rapid_sim_set_alert_set_device: borrowed)
    unsigned char* h_alert_stage = nullptr;  // pinned staging of rapid_sim_set_alert_set
    unsigned char* h_vstage = nullptr;       // pinned staging of a view change's uploads (member flags, the nodes that left / came, new
    size_t vstage_bytes = 0;                 // NodeIds): sized from n_max by presize_view, so that rebuild_view never waits for a copy
    size_t alert_stage_bytes = 0;
    bool vstage_busy = false;  // asynchronous copies out of h_vstage may still be in flight (cleared when a view change's last kernel has answered)
    hipEvent_t ev_alert = nullptr;
    hipStream_t stream_gen = nullptr;  // tiled rounds: the next tile's deliveries are made here while this tile is tallied
    hipEvent_t ev_gen_ready[2] = {nullptr, nullptr}, ev_gen_free[2] = {nullptr, nullptr}, ev_gen_inputs = nullptr;
    bool alert_copy_pending = false;
    long long n_alert_set = -1;
    bool trusted = false, all_down = false;
    bool all_current = false;  // every alert the round index saw carries the engine's configuration id
    bool trust_copies = false;  // the caller vouches that the deliveries are copies of the declared alerts (rapid_sim_trust_alert_copies)
    bool no_late_copies = false;     // ... and (level 2) that none of them carries another configuration id: no late deliveries among them
    bool streams_generated = false;  // the delivered records were laid down by the library itself, from the declared alerts
    DevBuf<unsigned int> d_loadflags;
    DevBuf<unsigned int> d_adj;
    hipEvent_t ev_idx0 = nullptr, ev_idx1 = nullptr;  // around the round-index kernels; read lazily (rapid_sim_index_info)
    bool index_ms_pending = false;
    bool time_index = false;  // somebody has asked for the index build's device time: the NEXT build is bracketed by events (two more
                              // packets in the queue of every round otherwise, for a number nobody reads)
    bool vote_lds_attr_set = false;
    const int* idxwork_clean_at = nullptr;  // the index work area is known to be all zero for this allocation and node count
    int idxwork_clean_n = -1;
    bool tally_bitmaps_valid = false;  // d_bitmaps holds the voters' proposals of the last tally launch as slot bitmaps (TallyParams::bitmaps)
    bool tally_votes_valid = false;  // d_voteback holds the vote statistics of the last tally launch (tally_kernel.h: vote_res)
    // the fast round settled inside the tally launch (tally_kernel.h: TallyParams::vote_cand): d_voteback holds the COMPLETE answer block
    // of the last launch -- candidate, its verified votes, voters, its node list -- and, for a population held by one rank, the
    // host-mapped page received it under sequence number tally_settled_seq; no counting or verifying launch follows
    bool tally_settled_valid = false;
    unsigned int tally_settled_seq = 0;
    bool settle_in_tally = true;  // (false inside a tiled round: its votes are accumulated across the tiles' launches by kernels of their own)
    DevBuf<unsigned long long> d_vote_cand;   // [4 + 64]
    DevBuf<unsigned int> d_vote_deferred;     // [1 + kVoteDeferredCap]
    bool stats_fresh = false;  // the statistics were zeroed by the index build of this very call
    hipEvent_t ev0 = nullptr, ev1 = nullptr;  // timing events, created once (no create / destroy per call, nothing to leak on an error path)
    DevBuf<unsigned short> d_dict, d_decl, d_adj_off, d_trank;
    DevBuf<unsigned int> d_tbits, d_tent;  // compressed dictionary (index_build_block_kernel)
    DevBuf<unsigned int> d_entries;        // dict_entry per node for rounds whose tables stay in memory
    // the hashed dictionary of packed rounds (index_hash_kernel): bucket bounds, remainders, member bits; the renumbered per-slot tables
    DevBuf<unsigned short> d_hoff, d_hnew, d_smask2;
    DevBuf<unsigned int> d_hrem, d_hmem, d_adj2;  // (d_hrem: bytes, allocated in whole dwords -- the tally stages it dword by dword)
    DevBuf<int> d_nos2;
    unsigned int hash_mul = 0;
    bool hash_lds_attr_set = false;
    DevBuf<uint2> d_gen_res;          // rapid_sim_generate: the alert set resolved once ({entry, core word} per alert)
    DevBuf<unsigned int> d_gen_keep;  // ... per-batch delivery thresholds
    DevBuf<long long> d_gen_boff;
    DevBuf<uint4> d_gen_bat;          // ... {first alert, length, the first alert's resolved record} per batch (gen_pack_batches_kernel)
    DevBuf<int> d_gen_rx;
    float generate_ms = 0.f;
    int n_touched = 0;
    int dict_mode = 3;  // rapid::kDictDirect / kDictCompressed / kDictMemory / kDictHashed for boundary records, kDictResolved for generated resident ones
    bool lds_attr_set = false;
    bool packed = false;  // the round's detector state: two slots per LDS word (tally_kernel.h: PackedSlotDetector)
    DevBuf<unsigned int> d_errflags;  // sticky per loaded stream set: bit0 = a delivered report is not covered by the declared alert set
    DevBuf<int> d_idxblk;  // per-workgroup hot / touched counts of the chunked index build
    DevBuf<int> d_node_of_slot, d_idxwork;  // d_idxwork = gmask[N] | info[8] of the round index build
    int n_slots = 0, n_hot = 0, n_adj = 0;
    float index_ms = 0.f;
    // launch geometry chosen from the index
    int waves_per_block = 1, grid_blocks = 1, lds_bytes = 0;
    bool tables_in_lds = true;
    int num_cus = 256;

    // ---- a round taken tile by tile (rapid_sim_round_tiled) ----
    long long out_base = 0;      // receiver offset of the running tile in the per-receiver result arrays (0 outside a tiled round)
    int tiled_total = 0;         // receivers of the last tiled round (its per-receiver results cover all of them), 0: none
    const unsigned long long* tiled_block = nullptr;  // ... and the answer block its last pass left for the all-gather (d_gather)
    int tiled_last_base = 0, tiled_last_n = 0;  // the tile whose proposals are still resident (rapid_sim_proposal)
    DevBuf<unsigned long long> d_vacc;  // vote accumulator: acc[8] | candidate bitmap | candidate list (vote_kernels.h)
    double tiled_ms[4] = {0, 0, 0, 0};  // the last tiled round: total wall, tiles, passes, records delivered / 1e6

    // ---- votes ----
    DevBuf<unsigned long long> d_hist, d_winner, d_mm, d_mismatch, d_voteback, d_gather;
    DevBuf<int> d_ref;
    std::vector<int> decided_cut;  // ring-0 order
    bool have_decision = false;
    unsigned long long* h_pinned = nullptr;  // pinned staging for the vote read-back (sharded populations)
    // host-mapped mailbox the kernels write their small answers into (no copy enqueued, the host reads it after a
    // synchronisation): [0, 64) the round index's info[8], [64, ...) the vote count's res[] + representative list
    bool idx_lds_attr_set = false;
    unsigned char* h_arena = nullptr;  // the engine's one pinned, host-mapped block (ensure_arena): mailbox | view staging
    size_t arena_bytes = 0;
    unsigned char* h_mail = nullptr;
    unsigned int mail_seq = 0;  // sequence numbers of the answers polled from the mailbox
    unsigned char* d_mail = nullptr;
    size_t mail_bytes = 0;

    // ---- multi-GPU ----
    ncclComm_t comm = nullptr;
    int rank = 0, n_ranks = 1;
};

struct rapid_cd {
    rapid_engine* eng = nullptr;
    int K = 0, H = 0, L = 0;
    int n_nodes = 0;
    DevBuf<unsigned short> d_state;
    DevBuf<int> d_scal, d_out, d_counts, d_out_n;
    DevBuf<unsigned char> d_alerts;
};

namespace {

int fail(rapid_engine* h, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) h->err = buf;
    return code;
}

#define HIPCHK(h, call)                                                                          \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess)                                                                    \
            return fail((h), RAPID_EDEVICE, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                                     \
    } while (0)

#define NCCLCHK(h, call)                                                                          \
    do {                                                                                          \
        ncclResult_t r_ = (call);                                                                 \
        if (r_ != ncclSuccess)                                                                    \
            return fail((h), RAPID_EDEVICE, "%s failed: %s (%s:%d)", #call, ncclGetErrorString(r_), \
                        __FILE__, __LINE__);                                                      \
    } while (0)

// After every kernel launch of the view path: a launch the runtime refused (bad geometry, LDS request over the limit) is reported
// by the kernel's name, not by whatever call comes across the sticky error later.  Test build with RAPID_SYNC_LAUNCHES set: the stream
// is synchronised as well, so a kernel that FAULTS is named too (the product never synchronises here).
#define LAUNCHCHK(h, name)                                                                                           \
    do {                                                                                                             \
        hipError_t e_ = hipGetLastError();                                                                           \
        if (e_ == hipSuccess && env_knob("RAPID_SYNC_LAUNCHES")) e_ = hipStreamSynchronize((h)->stream);             \
        if (e_ != hipSuccess)                                                                                        \
            return fail((h), RAPID_EDEVICE, "kernel %s: %s (%s:%d)", name, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

bool valid_khl(int K, int H, int L) {
    // R/MultiNodeCutDetector.java:52 -- if (H > K || L > H || K < K_MIN || L <= 0 || H <= 0) throw
    return !(H > K || L > H || K < 3 || L <= 0 || H <= 0);
}

inline unsigned grid_for(long long n, int block) { return (unsigned)((n + block - 1) / block); }

int total_records_fwd(rapid_engine* h, long long* n);  // (defined with the stream-loading calls below)

int use_device(rapid_engine* h) {
    HIPCHK(h, hipSetDevice(h->cfg.device_id));
    return RAPID_OK;
}

int ensure_mailbox(rapid_engine* h);
int ensure_arena(rapid_engine* h);

// A BORROWED device buffer (rapid_sim_attach_streams_device, rapid_sim_load_streams_device, rapid_sim_set_alert_set_device) is read
// by kernels long after the call that handed it over: a host pointer, a pointer into another device's memory or a length that runs
// past the allocation would surface as a GPU memory fault -- which ends the process (in a deployment: the JVM) -- instead of as an
// error code.  So the runtime is asked what the pointer is: memory this engine's device can read (its own allocations, or
// host-mapped / managed memory), with [p, p + bytes) inside ONE allocation.  ~1 us per pointer, host side only; no launch, no wait.
int check_borrowed(rapid_engine* h, const void* p, unsigned long long bytes, const char* what) {
    if (p == nullptr) return bytes == 0 ? RAPID_OK : fail(h, RAPID_EINVAL, "%s: null pointer with %llu bytes", what, bytes);
    hipPointerAttribute_t at;
    std::memset(&at, 0, sizeof at);
    const hipError_t e = hipPointerGetAttributes(&at, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(h, RAPID_EINVAL, "%s: %p is not memory the HIP runtime knows (%s): a device pointer is required", what, p, hipGetErrorString(e));
    }
    if (at.type == hipMemoryTypeUnregistered)
        return fail(h, RAPID_EINVAL, "%s: %p is ordinary host memory: a device pointer is required", what, p);
    if (at.type == hipMemoryTypeDevice && at.device != h->cfg.device_id)
        return fail(h, RAPID_EINVAL, "%s: %p lives on device %d, the engine runs on device %d", what, p, at.device, h->cfg.device_id);
    if (bytes) {
        hipDeviceptr_t base = nullptr;
        size_t size = 0;
        if (hipMemGetAddressRange(&base, &size, const_cast<void*>(p)) == hipSuccess) {
            const unsigned long long off = (unsigned long long)(reinterpret_cast<uintptr_t>(p) - reinterpret_cast<uintptr_t>(base));
            if (off > size || bytes > size - off)
                return fail(h, RAPID_EINVAL, "%s: %llu bytes at %p run past the end of their allocation (%zu bytes from %p)", what, bytes, p, size, (void*)base);
        } else {
            (void)hipGetLastError();  // (a kind of memory without a range record, e.g. registered host memory: the type check above stands)
        }
    }
    return RAPID_OK;
}

// Rebuilds rings, tables, state template and configuration id from the host member flags.
// Is any of `ids` (sorted, distinct) among the identifiers seen so far?  A binary search per id in the device's sorted copy.
static int ids_seen_any(rapid_engine* h, const std::vector<std::pair<int64_t, int64_t>>& ids, bool* any) {
    *any = false;
    if (ids.empty() || h->n_ids_dev == 0) return RAPID_OK;
    const size_t nn = ids.size();
    std::vector<long long> flat(2 * nn);
    for (size_t i = 0; i < nn; ++i) {
        flat[i] = ids[i].first;
        flat[nn + i] = ids[i].second;
    }
    HIPCHK(h, h->d_ids_new.ensure(2 * nn));
    if (nn <= 16384 && h->h_vstage != nullptr && h->vstage_bytes >= 16 * nn && !h->vstage_busy) {
        // a cut's NodeIds: through the pinned staging block, answered into the mailbox (word 13 = 2 x the call's sequence number +
        // the answer): no pageable copy, no stream synchronisation -- this check stands in front of every view change
        if (int rc = ensure_mailbox(h)) return rc;
        std::memcpy(h->h_vstage, flat.data(), 16 * nn);
        h->vstage_busy = true;
        HIPCHK(h, hipMemcpyAsync(h->d_ids_new.p, h->h_vstage, 16 * nn, hipMemcpyHostToDevice, h->stream));
        const unsigned int seq = ++h->mail_seq & 0x3FFFFFFFu;
        HIPCHK(h, h->d_idacc.ensure(2));
        if (!h->idacc_clean) {
            HIPCHK(h, hipMemsetAsync(h->d_idacc.p, 0, 8, h->stream));
            h->idacc_clean = true;  // (the kernel's last workgroup leaves both words zero)
        }
        hipLaunchKernelGGL(rapid::ids_contains_publish_kernel, dim3(grid_for((long long)nn, 256)), dim3(256), 0, h->stream, h->d_ids_hi.p, h->d_ids_lo.p,
                           h->n_ids_dev, h->d_ids_new.p, h->d_ids_new.p + nn, (int)nn, h->d_idacc.p, reinterpret_cast<volatile unsigned int*>(h->d_mail), 13, seq);
        HIPCHK(h, hipGetLastError());
        volatile unsigned int* const w = reinterpret_cast<volatile unsigned int*>(h->h_mail) + 13;
        const auto t0 = std::chrono::steady_clock::now();
        while ((*w >> 1) != seq) {
            if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) {  // (a faulted kernel never answers)
                HIPCHK(h, hipStreamSynchronize(h->stream));
                HIPCHK(h, hipGetLastError());
                if ((*w >> 1) != seq) return fail(h, RAPID_EDEVICE, "kernel finished without publishing its answer");
            }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        h->vstage_busy = false;  // (the kernel that answered ran behind the copy)
        *any = (*w & 1u) != 0u;
        return RAPID_OK;
    }
    HIPCHK(h, h->d_loadflags.ensure(2));
    HIPCHK(h, hipMemsetAsync(h->d_loadflags.p, 0, 8, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_ids_new.p, flat.data(), 16 * nn, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(rapid::ids_contains_kernel, dim3(grid_for((long long)nn, 256)), dim3(256), 0, h->stream, h->d_ids_hi.p, h->d_ids_lo.p, h->n_ids_dev,
                       h->d_ids_new.p, h->d_ids_new.p + nn, (int)nn, h->d_loadflags.p);
    unsigned int f = 0;
    HIPCHK(h, hipMemcpyAsync(&f, h->d_loadflags.p, 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    *any = f != 0u;
    return RAPID_OK;
}

// All K rings of `count` (sortable key, node) pairs per ring (ring k = [k count, (k + 1) count)), sorted by (key, node) -- the order a
// stable sort by key gives members handed over in ascending node order.
//   count <= kJoinSortMax (8,192: every small view, BASELINE configs[0] and [1]): the engine's own two kernels -- runs sorted in LDS,
//   then every pair finds its place among the other runs (view_kernels.h) -- no library, no temporary storage, no scratch memory;
//   beyond: the library's segmented radix sort with a configuration whose kernels keep their items in registers (RingSortConfig).
// Until round 5 every build went through the library's default configuration, whose gfx950 kernels spill 148 bytes per lane into
// scratch memory -- the only kernels of the product that needed the runtime to set up a scratch arena on the queue, and they ran inside
// the first call every user makes; tests/test_build.py now pins "no kernel of the product uses scratch".
using RingSortConfig = rocprim::segmented_radix_sort_config<8, rocprim::kernel_config<256, 8>, rocprim::DisabledWarpSortConfig, false>;

static int sort_rings(rapid_engine* h, unsigned long long* keys_in, unsigned long long* keys_out, int* vals_in, int* vals_out, int count) {
    const int K = h->cfg.K;
    hipStream_t st = h->stream;
    if (count <= rapid::kJoinSortMax) {
        const long long kc = (long long)K * count;
        hipLaunchKernelGGL(rapid::ring_sort_runs_kernel, dim3((unsigned)(K * ((count + rapid::kJoinRun - 1) / rapid::kJoinRun))), dim3(rapid::kJoinRun / 2), 0, st,
                           keys_in, vals_in, count);
        LAUNCHCHK(h, "ring_sort_runs_kernel");
        hipLaunchKernelGGL(rapid::ring_merge_runs_kernel, dim3(grid_for(kc, 256)), dim3(256), 0, st, keys_in, vals_in, count, K, keys_out, vals_out);
        LAUNCHCHK(h, "ring_merge_runs_kernel");
        return RAPID_OK;
    }
    h->seg_host.resize((size_t)K + 1);
    for (int k = 0; k <= K; ++k) h->seg_host[(size_t)k] = k * count;
    HIPCHK(h, h->d_seg_off.ensure((size_t)K + 1));
    HIPCHK(h, hipMemcpyAsync(h->d_seg_off.p, h->seg_host.data(), sizeof(int) * ((size_t)K + 1), hipMemcpyHostToDevice, st));
    size_t tmp_bytes = 0;
    HIPCHK(h, rocprim::segmented_radix_sort_pairs<RingSortConfig>(nullptr, tmp_bytes, keys_in, keys_out, vals_in, vals_out, (unsigned int)((size_t)K * count),
                                                                  (unsigned int)K, h->d_seg_off.p, h->d_seg_off.p + 1, 0, 64, st));
    HIPCHK(h, h->d_sort_tmp.ensure(tmp_bytes + 16));
    HIPCHK(h, rocprim::segmented_radix_sort_pairs<RingSortConfig>(h->d_sort_tmp.p, tmp_bytes, keys_in, keys_out, vals_in, vals_out,
                                                                  (unsigned int)((size_t)K * count), (unsigned int)K, h->d_seg_off.p, h->d_seg_off.p + 1, 0, 64, st));
    return RAPID_OK;
}

// Every buffer a view change can need, at the size the engine's capacity (n_max) allows: a view change -- on the path from a
// decided cut to the next configuration id -- then never allocates or frees device memory (an allocation inside
// rapid_apply_cut was measured at up to 190 ms against 4.5 ms for the change itself).  identifiersSeen is the exception: it
// is never pruned (R/MembershipView.java:167-201) and grows past any bound in the end; it starts at twice the capacity.
int ensure_mailbox(rapid_engine* h);
static int await_mail(rapid_engine* h, int word_index, unsigned int want);

int presize_view(rapid_engine* h) {
    const size_t K = (size_t)h->cfg.K, N = (size_t)h->cfg.n_max, km = K * N;
    const size_t J = N / 4 + 2, kj = K * J;  // a change with more joiners than a quarter of the view sorts afresh (rebuild_view)
    HIPCHK(h, h->d_member.ensure(N));
    HIPCHK(h, h->d_members.ensure(N));
    HIPCHK(h, h->d_sort_keys.ensure(km));
    HIPCHK(h, h->d_sort_vals.ensure(km));
    HIPCHK(h, h->d_ring_skeys.ensure(km));
    HIPCHK(h, h->d_ring.ensure(km));
    HIPCHK(h, h->d_pos.ensure(km));
    HIPCHK(h, h->d_obs.ensure(km));
    HIPCHK(h, h->d_subj.ensure(km));
    HIPCHK(h, h->d_cfg_out.ensure(1));
    HIPCHK(h, h->d_chunk_kept.ensure(K * ((N + rapid::kRingChunk - 1) / rapid::kRingChunk + 1)));
    HIPCHK(h, h->d_chunk_base.ensure(K * ((N + rapid::kRingChunk - 1) / rapid::kRingChunk + 2)));
    HIPCHK(h, h->d_chunk_lb.ensure(K * ((N + rapid::kRingChunk - 1) / rapid::kRingChunk + 2)));
    HIPCHK(h, h->d_nonmembers.ensure(N + 1));
    HIPCHK(h, h->d_joiners.ensure(J));
    HIPCHK(h, h->d_join_keys.ensure(kj));
    HIPCHK(h, h->d_join_skeys.ensure(kj));
    HIPCHK(h, h->d_join_vals.ensure(kj));
    HIPCHK(h, h->d_join_nodes.ensure(kj));
    HIPCHK(h, h->d_seg_off.ensure(K + 1));
    HIPCHK(h, h->d_ids_new.ensure(2 * N));
    HIPCHK(h, h->d_ids_hi.ensure(2 * N));
    HIPCHK(h, h->d_ids_lo.ensure(2 * N));
    HIPCHK(h, h->d_ids_hi2.ensure(2 * N));
    HIPCHK(h, h->d_ids_lo2.ensure(2 * N));
    HIPCHK(h, h->d_cfg_partial.ensure(2 * 512));
    HIPCHK(h, h->d_loadflags.ensure(2));
    HIPCHK(h, h->d_q4_rows.ensure(km));
    HIPCHK(h, h->d_q4_valid.ensure(N));
    HIPCHK(h, h->d_q4_nodes.ensure(N));
    HIPCHK(h, h->d_gone.ensure(N));
    // the library sort's scratch: for all K rings at full size, and for the joiners of one change
    size_t tmp_full = 0, tmp_join = 0;
    HIPCHK(h, rocprim::segmented_radix_sort_pairs<RingSortConfig>(nullptr, tmp_full, h->d_sort_keys.p, h->d_ring_skeys.p, h->d_sort_vals.p, h->d_ring.p, (unsigned int)km,
                                                  (unsigned int)K, h->d_seg_off.p, h->d_seg_off.p + 1, 0, 64, h->stream));
    HIPCHK(h, rocprim::segmented_radix_sort_pairs<RingSortConfig>(nullptr, tmp_join, h->d_join_keys.p, h->d_join_skeys.p, h->d_join_vals.p, h->d_join_nodes.p, (unsigned int)kj,
                                                  (unsigned int)K, h->d_seg_off.p, h->d_seg_off.p + 1, 0, 64, h->stream));
    HIPCHK(h, h->d_sort_tmp.ensure(std::max(tmp_full, tmp_join) + 16));
    return RAPID_OK;
}

#ifdef RAPID_TEST_BUILD
// RAPID_DEBUG_ADDR: where every buffer of the engine lives, against pthread_self() (the round-5 fault report names one address and the
// main thread's: with the same binaries the distances between mappings repeat from box to box)
void dump_addresses(rapid_engine* h) {
    const unsigned long long self = (unsigned long long)pthread_self();
    fprintf(stderr, "ADDR pthread_self %llx\n", self);
#define RAPID_DUMP(b) fprintf(stderr, "ADDR %-14s %14llx bytes %10zu self-minus %llx\n", #b, (unsigned long long)(uintptr_t)h->b.p, h->b.cap * sizeof(*h->b.p), self - (unsigned long long)(uintptr_t)h->b.p)
    RAPID_DUMP(d_blob); RAPID_DUMP(d_host_off); RAPID_DUMP(d_ports); RAPID_DUMP(d_keys); RAPID_DUMP(d_hx_host0); RAPID_DUMP(d_hx_port0);
    RAPID_DUMP(d_member); RAPID_DUMP(d_members); RAPID_DUMP(d_sort_keys); RAPID_DUMP(d_sort_vals); RAPID_DUMP(d_ring_skeys); RAPID_DUMP(d_ring);
    RAPID_DUMP(d_pos); RAPID_DUMP(d_obs); RAPID_DUMP(d_subj); RAPID_DUMP(d_ids_hi); RAPID_DUMP(d_ids_lo); RAPID_DUMP(d_ids_hi2); RAPID_DUMP(d_ids_lo2);
    RAPID_DUMP(d_ids_new); RAPID_DUMP(d_cfg_out); RAPID_DUMP(d_cfg_partial); RAPID_DUMP(d_chunk_kept); RAPID_DUMP(d_chunk_base); RAPID_DUMP(d_chunk_lb);
    RAPID_DUMP(d_nonmembers); RAPID_DUMP(d_joiners); RAPID_DUMP(d_join_keys); RAPID_DUMP(d_join_skeys); RAPID_DUMP(d_join_vals); RAPID_DUMP(d_join_nodes);
    RAPID_DUMP(d_seg_off); RAPID_DUMP(d_sort_tmp); RAPID_DUMP(d_loadflags); RAPID_DUMP(d_q4_rows); RAPID_DUMP(d_q4_valid); RAPID_DUMP(d_q4_nodes); RAPID_DUMP(d_gone);
#undef RAPID_DUMP
    fprintf(stderr, "ADDR %-14s %14llx bytes %10zu self-minus %llx\n", "h_mail", (unsigned long long)(uintptr_t)h->h_mail, h->mail_bytes, self - (unsigned long long)(uintptr_t)h->h_mail);
    fprintf(stderr, "ADDR %-14s %14llx bytes %10zu self-minus %llx\n", "h_vstage", (unsigned long long)(uintptr_t)h->h_vstage, h->vstage_bytes, self - (unsigned long long)(uintptr_t)h->h_vstage);
}
#endif

int rebuild_view(rapid_engine* h) {
    const int K = h->cfg.K, N = h->n_nodes;
    hipStream_t st = h->stream;
    const bool timing = env_knob("RAPID_TIME_VIEW") != nullptr;  // profiling knob: where a view change spends its time (stderr)
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!timing) return;
        (void)hipStreamSynchronize(st);
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "rebuild_view %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    // What changed since the rings on the device were built: members that left, members that came (R/MembershipService.java:
    // 385-430 applies a decided cut as a sequence of ringDelete / ringAdd).  Everybody else keeps its place in every ring.
    const bool have_rings = h->ring_m > 0 && h->ring_member.size() == (size_t)N;
    std::vector<int> joiners, gone;
    int removed = 0, M = 0;
    const bool by_list = have_rings && h->changed_valid;  // (the device's member flags are then the ones the rings were built from)
    if (by_list) {
        // (in the order the cut names them: nothing below depends on the order of these two lists -- the joiners are sorted per ring
        // by their keys on the device, the rest are per-node updates -- and sorting 15,000 node indices here was a third of the host's
        // share of a view change at 10^6 members; a node the cut names twice is taken once)
        if (h->change_stamp.size() < (size_t)N) h->change_stamp.resize((size_t)N, 0u);
        if (++h->change_epoch == 0u) {
            std::fill(h->change_stamp.begin(), h->change_stamp.end(), 0u);
            h->change_epoch = 1u;
        }
        for (const int n : h->changed) {
            if (h->change_stamp[(size_t)n] == h->change_epoch) continue;
            h->change_stamp[(size_t)n] = h->change_epoch;
            const bool now = h->member[(size_t)n] != 0, was = h->ring_member[(size_t)n] != 0;
            if (now && !was) joiners.push_back(n);
            if (was && !now) gone.push_back(n);
        }
        removed = (int)gone.size();
        M = h->ring_m - removed + (int)joiners.size();
    } else {
        for (int n = 0; n < N; ++n) {
            const bool now = h->member[(size_t)n] != 0;
            M += now ? 1 : 0;
            if (have_rings) {
                const bool was = h->ring_member[(size_t)n] != 0;
                if (now && !was) joiners.push_back(n);
                if (was && !now) {
                    ++removed;
                    gone.push_back(n);
                }
            }
        }
    }
    // Q4: ringDelete drops the memoised observers of the node itself and of its ring predecessors (TreeSet.lower, no
    // wrap-around) -- read off the tables of the view that is about to change; ringAdd those of the joiner's new predecessors
    // (further down, off the new tables)
    // Everything this change uploads goes through the pinned staging block (presize_view: member flags | nodes that left | nodes
    // that came | members | new NodeIds): the copies are asynchronous, the block lives as long as the engine, and the one wait of
    // a view change is for the configuration id at its end.  (Each upload used to come out of a local vector and was followed
    // by a stream synchronisation so that the vector could go: five waits of ~20-40 us around kernels of a few us each.)
    if (!h->h_vstage || h->vstage_bytes < (size_t)29 * (size_t)N + 256) return fail(h, RAPID_ESTATE, "view buffers are not sized (rapid_view_build first)");
    // The staging block is written with plain memcpy: the previous view change's copies out of it are known to be done when that
    // change got as far as its configuration id.  One that returned early (a device error on the way) may have left copies in
    // flight: waited for here, on that rare path only.
    if (h->vstage_busy) {
        HIPCHK(h, hipStreamSynchronize(st));
        h->vstage_busy = false;
    }
    h->vstage_busy = true;
    const size_t a16 = 15, n4 = ((size_t)N * 4 + a16) & ~a16;  // (bytes of N ints, rounded up to 16)
    unsigned char* const stage_member = h->h_vstage;
    unsigned char* const stage_lists = h->h_vstage + (((size_t)N + a16) & ~a16);
    int* const stage_gone = reinterpret_cast<int*>(stage_lists);
    int* const stage_join = reinterpret_cast<int*>(stage_lists + n4);
    int* const stage_members = reinterpret_cast<int*>(stage_lists + 2 * n4);
    long long* const stage_ids = reinterpret_cast<long long*>(stage_lists + 3 * n4);  // 16 N bytes
    // the nodes that left and, right behind them, the nodes that came: one list on the device (ring_patch_kernel takes it whole)
    std::memcpy(stage_gone, gone.data(), sizeof(int) * gone.size());
    std::memcpy(stage_gone + gone.size(), joiners.data(), sizeof(int) * joiners.size());
    (void)stage_join;
    bool gone_cleared = false;  // the member flags of the nodes that left were cleared by the kernel that drops their memo entries
    // (ring_now / m_now: the rings of the view the predecessors are taken from -- the old ones for the nodes that leave, the new ones
    // for the joiners)
    auto q4_drop = [&](const int* d_nodes, size_t m, int self, bool clear_members, const int* ring_now, int m_now) -> int {
        if (m == 0 || !h->d_subj.p || !ring_now || !h->d_q4_valid.p) return RAPID_OK;
        hipLaunchKernelGGL(rapid::q4_invalidate_kernel, dim3(grid_for((long long)m * K, 256)), dim3(256), 0, st, h->d_subj.p, ring_now, m_now, d_nodes,
                           (int)m, N, K, h->d_q4_valid.p, self, clear_members ? h->d_member.p : (unsigned char*)nullptr);
        LAUNCHCHK(h, "q4_invalidate_kernel");
        gone_cleared = gone_cleared || clear_members;
        return RAPID_OK;
    };
    const int J = (int)joiners.size();
    const int* d_joined = nullptr;  // the joiners on the device: behind the nodes that left, in the same buffer
    if (have_rings && (!gone.empty() || J > 0)) {
        HIPCHK(h, h->d_gone.ensure(gone.size() + (size_t)J));
        HIPCHK(h, hipMemcpyAsync(h->d_gone.p, stage_gone, sizeof(int) * (gone.size() + (size_t)J), hipMemcpyHostToDevice, st));
        d_joined = h->d_gone.p + gone.size();
        if (!gone.empty()) {
            int rc = q4_drop(h->d_gone.p, gone.size(), 1, by_list && h->d_member.p != nullptr, h->d_ring.p, h->ring_m);  // (:181-195)
            if (rc) return rc;
        }
    }
    h->n_members = M;
    lap("host scan + q4");

    HIPCHK(h, h->d_member.ensure((size_t)N));
    if (by_list) {  // the flags on the device are patched where they changed
        const int n_clear = gone_cleared ? 0 : removed;
        if (n_clear > 0 || J > 0) {
            hipLaunchKernelGGL(rapid::member_patch_kernel, dim3(grid_for((long long)std::max(n_clear, J), 256)), dim3(256), 0, st, h->d_member.p,
                               n_clear > 0 ? h->d_gone.p : (const int*)nullptr, n_clear, J > 0 ? d_joined : (const int*)nullptr, J);
            LAUNCHCHK(h, "member_patch_kernel");
        }
    } else {
        std::memcpy(stage_member, h->member.data(), (size_t)N);
        HIPCHK(h, hipMemcpyAsync(h->d_member.p, stage_member, (size_t)N, hipMemcpyHostToDevice, st));
    }
    const size_t km = (size_t)K * (size_t)std::max(M, 1);
    HIPCHK(h, h->d_sort_keys.ensure(km));  // (DevBuf::ensure does not keep contents: the rings themselves are only grown where
    HIPCHK(h, h->d_sort_vals.ensure(km));  //  they are about to be written from scratch)
    HIPCHK(h, h->d_pos.ensure((size_t)K * N));
    HIPCHK(h, h->d_obs.ensure((size_t)K * N));
    HIPCHK(h, h->d_subj.ensure((size_t)K * N));
    HIPCHK(h, h->d_cfg_out.ensure(1));

    // incremental while the change is small against the view (a decided cut); a view that is mostly new is sorted afresh
    const bool incremental = have_rings && M > 0 && (long long)J * 4 <= (long long)h->ring_m && (h->force_exact & 16384) == 0;
    bool tables_patched = false;
    if (incremental && (removed > 0 || J > 0)) {
        // Old ring k minus the removed nodes, merged with the joiners in the order of their ring-k keys
        // (R/MembershipView.java:123-201: each TreeSet loses / gains the endpoint, nothing else moves): three launches over
        // chunks of the old rings instead of a sort of all K x M keys -- and the only thing a removals-only cut needs.
        const int m_old = h->ring_m;
        const int n_chunks = (m_old + rapid::kRingChunk - 1) / rapid::kRingChunk;
        HIPCHK(h, h->d_chunk_kept.ensure((size_t)K * n_chunks));
        hipLaunchKernelGGL(rapid::ring_count_kernel, dim3((unsigned)(K * n_chunks)), dim3(rapid::kRingChunk), 0, st, h->d_ring.p, m_old, n_chunks,
                           h->d_member.p, h->d_chunk_kept.p);
        LAUNCHCHK(h, "ring_count_kernel");
        if (J > 0) {
            const size_t kj = (size_t)K * J;
            HIPCHK(h, h->d_join_keys.ensure(kj));
            HIPCHK(h, h->d_join_skeys.ensure(kj));
            HIPCHK(h, h->d_join_vals.ensure(kj));
            HIPCHK(h, h->d_join_nodes.ensure(kj));
            hipLaunchKernelGGL(rapid::ring_gather_kernel, dim3(grid_for((long long)kj, 256)), dim3(256), 0, st, h->d_keys.p, d_joined, J, N, K,
                               h->d_join_keys.p, h->d_join_vals.p);
            LAUNCHCHK(h, "ring_gather_kernel");
            // (a cut's joiners, up to 8,192 of them: the engine's own two sort kernels -- the library's sort has 0.27 ms of fixed cost)
            int rc = sort_rings(h, h->d_join_keys.p, h->d_join_skeys.p, h->d_join_vals.p, h->d_join_nodes.p, J);
            if (rc) return rc;
        }
        HIPCHK(h, h->d_chunk_base.ensure((size_t)K * ((size_t)n_chunks + 1)));
        HIPCHK(h, h->d_chunk_lb.ensure((size_t)K * ((size_t)n_chunks + 1)));
        hipLaunchKernelGGL(rapid::ring_chunk_prep_kernel, dim3((unsigned)K), dim3(1024), 0, st, h->d_ring.p, h->d_ring_skeys.p, m_old, n_chunks, h->d_chunk_kept.p,
                           h->d_join_skeys.p, h->d_join_nodes.p, J, h->d_chunk_base.p, h->d_chunk_lb.p);
        LAUNCHCHK(h, "ring_chunk_prep_kernel");
        hipLaunchKernelGGL(rapid::ring_scatter_kernel, dim3((unsigned)(K * n_chunks)), dim3(rapid::kRingChunk), 0, st, h->d_ring.p, h->d_ring_skeys.p,
                           m_old, n_chunks, h->d_member.p, h->d_chunk_base.p, h->d_chunk_lb.p, h->d_join_skeys.p, h->d_join_nodes.p, J, h->d_sort_vals.p,
                           h->d_sort_keys.p, M);
        LAUNCHCHK(h, "ring_scatter_kernel");
        if (J > 0) {
            hipLaunchKernelGGL(rapid::ring_join_kernel, dim3(grid_for((long long)K * J * 64, 256)), dim3(256), 0, st, h->d_ring.p, h->d_ring_skeys.p,
                               m_old, n_chunks, h->d_member.p, h->d_chunk_base.p, h->d_join_skeys.p, h->d_join_nodes.p, J, K, h->d_sort_vals.p,
                               h->d_sort_keys.p, M);
            LAUNCHCHK(h, "ring_join_kernel");
        }
        std::swap(h->d_ring, h->d_sort_vals);
        std::swap(h->d_ring_skeys, h->d_sort_keys);
        lap("rings: compact + merge");
        // The tables follow the rings where the cut touched them: the old neighbours of the nodes that left, the joiners and their new
        // neighbours, on every ring (ring_patch_kernel), then the rows of the non-members (their expected observers).  Two full
        // passes over the K x M positions with scattered reads were 0.38 ms of a view change at 10^6 members.
        if (by_list && M >= 2 && m_old >= 2) {
            hipLaunchKernelGGL(rapid::ring_patch_kernel, dim3(grid_for((long long)(removed + J) * K, 256)), dim3(256), 0, st, h->d_gone.p, removed, J, h->d_ring.p,
                               h->d_ring_skeys.p, M, h->d_keys.p, h->d_member.p, N, K, h->d_obs.p, h->d_subj.p);
            LAUNCHCHK(h, "ring_patch_kernel");
            HIPCHK(h, h->d_nonmembers.ensure((size_t)N + 1));
            HIPCHK(h, hipMemsetAsync(h->d_nonmembers.p, 0, 4, st));
            hipLaunchKernelGGL(rapid::nonmember_list_kernel, dim3(grid_for((long long)N, 1024)), dim3(1024), 0, st, h->d_member.p, N, h->d_nonmembers.p);
            LAUNCHCHK(h, "nonmember_list_kernel");
            hipLaunchKernelGGL(rapid::ring_nonmember_rows_kernel, dim3((unsigned)std::min<long long>(4096, std::max<long long>(1, grid_for((long long)K * (N - M), 256)))),
                               dim3(256), 0, st, h->d_ring.p, h->d_ring_skeys.p, M, h->d_keys.p, h->d_nonmembers.p, N, K, h->d_obs.p, h->d_subj.p);
            LAUNCHCHK(h, "ring_nonmember_rows_kernel");
            tables_patched = true;
        }
    } else if (M && !(incremental && removed == 0 && J == 0)) {
        HIPCHK(h, h->d_ring_skeys.ensure(km));
        HIPCHK(h, h->d_ring.ensure(km));
        int at = 0;
        for (int n = 0; n < N; ++n)
            if (h->member[(size_t)n]) stage_members[at++] = n;
        HIPCHK(h, h->d_members.ensure((size_t)M));
        HIPCHK(h, hipMemcpyAsync(h->d_members.p, stage_members, sizeof(int) * M, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(rapid::ring_gather_kernel, dim3(grid_for((long long)K * M, 256)), dim3(256), 0, st, h->d_keys.p,
                           h->d_members.p, M, N, K, h->d_sort_keys.p, h->d_sort_vals.p);
        LAUNCHCHK(h, "ring_gather_kernel");
        int rc = sort_rings(h, h->d_sort_keys.p, h->d_ring_skeys.p, h->d_sort_vals.p, h->d_ring.p, M);
        if (rc) return rc;
    }
    if (!tables_patched) {  // (a build, a change that sorted the rings afresh: every row from the rings; d_pos is this pass's scratch)
        if (M) {
            hipLaunchKernelGGL(rapid::ring_tables_kernel, dim3(grid_for((long long)K * M, 256)), dim3(256), 0, st, h->d_ring.p,
                               h->d_ring_skeys.p, h->d_keys.p, h->d_member.p, N, M, K, h->d_pos.p, h->d_obs.p, h->d_subj.p, 0);
            LAUNCHCHK(h, "ring_tables_kernel");
        }
        hipLaunchKernelGGL(rapid::ring_tables_kernel, dim3(grid_for((long long)K * N, 256)), dim3(256), 0, st, h->d_ring.p,
                           h->d_ring_skeys.p, h->d_keys.p, h->d_member.p, N, M, K, h->d_pos.p, h->d_obs.p, h->d_subj.p, 1);
        LAUNCHCHK(h, "ring_tables_kernel");
    }

    lap("tables");
    if (have_rings && !joiners.empty()) {  // ringAdd drops the entries of the joiner's new ring predecessors (:143-152)
        int rc = q4_drop(d_joined, joiners.size(), 0, false, h->d_ring.p, M);
        if (rc) return rc;
    }

    // identifiersSeen, sorted, on the device: everything again after a build; a cut's few new NodeIds are merged in
    if (!h->ids_pending.empty()) {
        if (!std::is_sorted(h->ids_pending.begin(), h->ids_pending.end())) rapid::sort_node_ids(h->ids_pending);  // (a cut's arrive sorted)
        const size_t nn = h->ids_pending.size(), ni = (size_t)h->n_ids_dev + nn;
        std::vector<long long> overflow;  // (more new NodeIds than the capacity in nodes -- a fresh build registers extra ids: not staged)
        long long* flat = stage_ids;
        if (nn > (size_t)N) {
            overflow.resize(2 * nn);
            flat = overflow.data();
        }
        for (size_t i = 0; i < nn; ++i) {
            flat[i] = h->ids_pending[i].first;
            flat[nn + i] = h->ids_pending[i].second;
        }
        HIPCHK(h, h->d_ids_new.ensure(2 * nn));
        HIPCHK(h, h->d_ids_hi2.ensure(ni));
        HIPCHK(h, h->d_ids_lo2.ensure(ni));
        HIPCHK(h, hipMemcpyAsync(h->d_ids_new.p, flat, 16 * nn, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(rapid::ids_merge_kernel, dim3(grid_for((long long)ni, 256)), dim3(256), 0, st, h->d_ids_hi.p, h->d_ids_lo.p, h->n_ids_dev,
                           h->d_ids_new.p, h->d_ids_new.p + nn, (int)nn, h->d_ids_hi2.p, h->d_ids_lo2.p);
        LAUNCHCHK(h, "ids_merge_kernel");
        if (!overflow.empty()) HIPCHK(h, hipStreamSynchronize(st));  // (`overflow` goes out of scope)
        std::swap(h->d_ids_hi, h->d_ids_hi2);
        std::swap(h->d_ids_lo, h->d_ids_lo2);
        h->n_ids_dev = (int)ni;
        h->ids_pending.clear();
    }
    lap("identifiers");
    int rc_mail = 0;
    {
        const int T = 1024;
        const long long total = 2ll * h->n_ids_dev + 2ll * M;
        // (one workgroup for everything was tried for small views -- one launch instead of two -- and is slower: forty dependent
        // hashes per thread against ten)
        const int G = (int)std::max<long long>(1, std::min<long long>(512, total / 8192));
        HIPCHK(h, h->d_cfg_partial.ensure((size_t)2 * G));
        // The configuration id is written straight into the host-mapped page (bytes 32..39, sequence word 10) by whichever kernel
        // finishes it: the host polls for it like for a round's answers -- no copy, no stream synchronisation
        if ((rc_mail = ensure_mailbox(h))) return rc_mail;
        long long* const d_cfg = reinterpret_cast<long long*>(h->d_mail + 32);
        volatile unsigned int* const d_seq = reinterpret_cast<volatile unsigned int*>(h->d_mail) + 10;
        const unsigned int seq = ++h->mail_seq;
        hipLaunchKernelGGL(rapid::config_id_kernel, dim3((unsigned)G), dim3(T), (size_t)T * 16, st, h->d_ids_hi.p, h->d_ids_lo.p,
                           h->n_ids_dev, h->d_ring.p, M, h->d_hx_host0.p, h->d_hx_port0.p, d_cfg, h->d_cfg_partial.p, d_seq, seq);
        LAUNCHCHK(h, "config_id_kernel");
        if (G > 1) {
            hipLaunchKernelGGL(rapid::config_id_final_kernel, dim3(1), dim3(512), 0, st, h->d_cfg_partial.p, G, d_cfg, d_seq, seq);
            LAUNCHCHK(h, "config_id_final_kernel");
        }
        if ((rc_mail = await_mail(h, 10, seq))) return rc_mail;
    }
    h->vstage_busy = false;  // (the kernel that answered runs behind every copy of this change on the stream)
    long long cfg = 0;
    std::memcpy(&cfg, h->h_mail + 32, 8);
    lap("configuration id");
    h->config_id = cfg;
    if (by_list) {
        for (const int n : gone) h->ring_member[(size_t)n] = 0;
        for (const int n : joiners) h->ring_member[(size_t)n] = 1;
    } else {
        h->ring_member = h->member;
    }
    h->changed.clear();
    h->changed_valid = true;
    h->ring_m = M;
    h->host_tables_valid = false;
    h->tallied = false;
    h->have_decision = false;
    h->index_valid = false;
#ifdef RAPID_TEST_BUILD
    if (env_knob("RAPID_DEBUG_ADDR")) dump_addresses(h);
#endif
    return RAPID_OK;
}

// A view change that failed on the way (a device error inside rebuild_view) may have patched the device's member flags and
// swapped the rings while ring_member / ring_m still describe the old ones: nothing incremental may be built on that.  The next
// change sorts the rings afresh from the host's member flags; if the identifiers of the change were already merged on the device
// the view has to be built again (rapid_view_build) -- the set cannot be un-merged.
int rebuild_view(rapid_engine* h);
void view_change_failed(rapid_engine* h, int n_ids_before) {
    h->changed.clear();
    h->changed_valid = false;
    h->ring_member.clear();
    h->ring_m = 0;
    h->host_tables_valid = false;
    h->index_valid = false;
    h->tallied = false;
    h->have_decision = false;
    if (h->n_ids_dev != n_ids_before) {
        h->view_built = false;
        h->err += " (the identifiers of the failed change are merged already: build the view again)";
        return;
    }
    // The device's rings may be swapped and its tables patched for the cut that failed, under the old configuration id: nothing may
    // be answered from them.  The rolled-back membership is built afresh right away (a full sort, the !have_rings path); if the
    // device cannot do that either, there is no view until rapid_view_build.
    const std::string first = h->err;
    if (rebuild_view(h) != RAPID_OK) {
        h->view_built = false;
        h->ring_member.clear();
        h->ring_m = 0;
        h->changed_valid = false;
        h->err = first + " (and the view could not be rebuilt from the rolled-back membership: " + h->err + "; build the view again)";
    } else {
        h->err = first + " (the view was rebuilt from the rolled-back membership)";
    }
}

int ensure_host_tables(rapid_engine* h) {
    if (h->host_tables_valid) return RAPID_OK;
    const int K = h->cfg.K, N = h->n_nodes, M = h->n_members;
    h->h_obs.resize((size_t)K * N);
    h->h_subj.resize((size_t)K * N);
    h->h_ring.resize((size_t)K * std::max(M, 1));
    h->h_keys.resize((size_t)K * N);
    HIPCHK(h, hipMemcpyAsync(h->h_obs.data(), h->d_obs.p, sizeof(int) * K * N, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->h_subj.data(), h->d_subj.p, sizeof(int) * K * N, hipMemcpyDeviceToHost, h->stream));
    if (M)
        HIPCHK(h, hipMemcpyAsync(h->h_ring.data(), h->d_ring.p, sizeof(int) * (size_t)K * M, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->h_keys.data(), h->d_keys.p, sizeof(long long) * K * N, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->host_tables_valid = true;
    if (!h->host_keys0_valid) {  // (the ring-0 keys came along: no copy of their own later)
        h->h_keys0.assign(h->h_keys.begin(), h->h_keys.begin() + N);
        h->host_keys0_valid = true;
    }
    return RAPID_OK;
}

int ensure_host_keys0(rapid_engine* h) {
    if (h->host_keys0_valid) return RAPID_OK;
    const int N = h->n_nodes;
    h->h_keys0.resize((size_t)std::max(N, 1));
    HIPCHK(h, hipMemcpyAsync(h->h_keys0.data(), h->d_keys.p, sizeof(long long) * (size_t)N, hipMemcpyDeviceToHost, h->stream));  // row 0 of [K][n_nodes]
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->host_keys0_valid = true;
    return RAPID_OK;
}

int check_node(rapid_engine* h, int node) {
    if (!h->view_built) return fail(h, RAPID_ESTATE, "view not built");
    if (node < 0 || node >= h->n_nodes) return fail(h, RAPID_EINVAL, "node index %d out of range [0,%d)", node, h->n_nodes);
    return RAPID_OK;
}

int copy_list(rapid_engine* h, const int* src, int n, int32_t* out, int32_t cap, int32_t* n_out) {
    if (n_out) *n_out = n;
    if (n > cap) return fail(h, RAPID_ECAPACITY, "output needs %d entries, capacity %d", n, cap);
    for (int i = 0; i < n; ++i) out[i] = src[i];
    return RAPID_OK;
}

// ONE pinned, host-mapped block per engine, allocated by rapid_engine_create and freed by rapid_engine_destroy, never in between:
//   [0, mail_bytes)   the mailbox the kernels write their small answers into -- [0, 64) the round index's info[8] and the sequence
//                     words, [64, 64 + A) the vote answer, [.., + A) the staging of the copied answer (sharded populations);
//   [mail_bytes, ..)  the staging block of a view change's uploads: member flags (N) | nodes that left (4 N) | nodes that came (4 N)
//                     | members, for a sort from scratch (4 N) | new NodeIds (16 N), N = n_max (+ 256: each part starts on a
//                     16-byte boundary).
// At least 64 KiB and a multiple of it, with explicit flags (mapped, coherent).  Rounds 2-5 kept the mailbox and the staging
// blocks as separate small hipHostMalloc allocations (4 KiB each for a small engine), allocated lazily and re-allocated on
// growth; DESIGN.md section 8 has the history of the fault report that ended that.
int ensure_arena(rapid_engine* h) {
    if (h->h_arena) return RAPID_OK;
    const size_t answer = (10 * 8 + ((size_t)h->max_cut + 1) * sizeof(int) + 63) & ~(size_t)63;
    const size_t mail = (64 + 2 * answer + 4095) & ~(size_t)4095;
    const size_t stage = ((size_t)29 * (size_t)h->cfg.n_max + 256 + 4095) & ~(size_t)4095;
    const size_t need = (mail + stage + 65535) & ~(size_t)65535;
    HIPCHK(h, hipHostMalloc(reinterpret_cast<void**>(&h->h_arena), need, hipHostMallocMapped | hipHostMallocCoherent));
    unsigned char* dev = nullptr;
    HIPCHK(h, hipHostGetDevicePointer(reinterpret_cast<void**>(&dev), h->h_arena, 0));
    std::memset(h->h_arena, 0, need);
    h->arena_bytes = need;
    h->h_mail = h->h_arena;
    h->d_mail = dev;
    h->mail_bytes = mail;
    h->h_pinned = reinterpret_cast<unsigned long long*>(h->h_mail + 64 + answer);
    h->h_vstage = h->h_arena + mail;
    h->vstage_bytes = need - mail;
    return RAPID_OK;
}

int ensure_mailbox(rapid_engine* h) {
    return h->h_mail ? RAPID_OK : fail(h, RAPID_ESTATE, "the engine has no mailbox (rapid_engine_create allocates it)");
}

// Waits for a kernel's answer in the host-mapped mailbox: the kernel's last act is a system-scope fence and the write of
// `want` into the polled word (bytes 56..63 of the page), so the host need not go through the runtime's completion path
// (hipStreamSynchronize costs ~20 us of wake-up latency, twice per round).  Falls back to it after 2 ms without an answer --
// a faulted kernel never answers.
static int await_mail(rapid_engine* h, int word_index, unsigned int want) {
    volatile unsigned int* const w = reinterpret_cast<volatile unsigned int*>(h->h_mail) + word_index;
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        for (int i = 0; i < 64; ++i) {
            if (*w == want) {
                std::atomic_thread_fence(std::memory_order_acquire);
                return RAPID_OK;
            }
#if defined(__x86_64__) || defined(__i386__)
            __builtin_ia32_pause();
#endif
        }
        if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipGetLastError());
    if (*w != want) return fail(h, RAPID_EDEVICE, "kernel finished without publishing its answer");
    return RAPID_OK;
}

// Builds the per-round index (touched / hot subjects, slot dictionary, hot adjacency) for the loaded streams under
// the current view, and picks the launch geometry of the tally kernel.
// launch statistics [workgroups][8] + the pool words of launch_tally, sized for the largest grid a launch can have
static size_t stats_words(const rapid_engine* h) { return (size_t)8 * (size_t)std::max(h->num_cus, 1) * 4 + 8; }

bool tally_is_trusted(const rapid_engine* h);
int ensure_tally_attrs(rapid_engine* h);

int build_round_index(rapid_engine* h) {
    const int N = h->n_nodes, K = h->cfg.K, L = h->cfg.L;
    if (h->n_alert_set < 0 && h->offsets_on_device_only) {  // nothing declared: the pass over the delivered records needs their number
        long long n_rec = 0;
        if (int rc = total_records_fwd(h, &n_rec)) return rc;
    }
    hipStream_t st = h->stream;
    {
        const int rc = ensure_mailbox(h);
        if (rc) return rc;
    }
    HIPCHK(h, h->d_idxwork.ensure((size_t)N + 8));
    HIPCHK(h, h->d_dict.ensure((size_t)N + 8));  // (+ 8: the tally kernel stages them 16 bytes at a time)
    HIPCHK(h, h->d_decl.ensure((size_t)N + 40));  // (the index build reads it 64 bytes at a time)
    HIPCHK(h, h->d_node_of_slot.ensure((size_t)N));
    HIPCHK(h, h->d_entries.ensure((size_t)N + 8));  // (+ 8: staged 16 bytes at a time)
    HIPCHK(h, h->d_adj_off.ensure((size_t)N + 1 + 8));
    const int tent_cap = 16384;  // touched nodes the compressed dictionary can hold (64 KiB of LDS)
    HIPCHK(h, h->d_tbits.ensure((size_t)(N + 31) / 32 + 1));
    HIPCHK(h, h->d_trank.ensure((size_t)(N + 31) / 32 + 1));
    HIPCHK(h, h->d_tent.ensure((size_t)tent_cap));
    unsigned int* const d_gmask = reinterpret_cast<unsigned int*>(h->d_idxwork.p);
    int* const d_info = h->d_idxwork.p + (size_t)N;
    const bool timed = h->time_index;  // (asked for since the last build: this one is bracketed by events)
    h->time_index = false;
    if (timed) {
        if (!h->ev_idx0) HIPCHK(h, hipEventCreate(&h->ev_idx0));
        if (!h->ev_idx1) HIPCHK(h, hipEventCreate(&h->ev_idx1));
        HIPCHK(h, hipEventRecord(h->ev_idx0, st));
    }
    const int adj_cap = 65536;
    HIPCHK(h, h->d_adj.ensure((size_t)adj_cap + 1));
    // A declared alert set over a population whose per-node tables fit one workgroup's LDS: the whole index in ONE launch
    // (index_kernels.h: index_fused_kernel -- the alert set read once, everything else computed in LDS, the tables written out);
    // testing knob bit 17: the two-kernel form (touch + one workgroup) whatever the size, as a cross-check.
    const bool chunked = N >= rapid::kIndexChunkedMin || (h->force_exact & 4096) != 0;
    const bool fused = h->n_alert_set >= 0 && N <= rapid::kIndexFusedMaxNodes && !chunked && (h->force_exact & 131072) == 0;
    if (fused) {
        if (!h->idx_lds_attr_set) {
            HIPCHK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(rapid::index_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                          rapid::index_fused_lds_bytes(rapid::kIndexFusedMaxNodes)));  // (next to its few static words: below the CU's 160 KiB)
            h->idx_lds_attr_set = true;
        }
        hipLaunchKernelGGL(rapid::index_fused_kernel, dim3(1), dim3(1024), (size_t)rapid::index_fused_lds_bytes(N), st, h->d_alerts, (long long)h->n_alert_set,
                           (long long)h->config_id, h->d_member.p, h->d_obs.p, N, K, L, h->d_dict.p, h->d_decl.p, h->d_node_of_slot.p, h->d_adj_off.p,
                           h->d_adj.p, adj_cap, h->d_tbits.p, h->d_trank.p, h->d_tent.p, tent_cap, reinterpret_cast<volatile int*>(h->d_mail),
                           (h->force_exact & (128 | 256 | 8192)) != 0 ? -1 : 160 * 1024 - rapid::kBlockStatsBytes, h->d_stats.p, (int)stats_words(h),
                           h->d_errflags.p, (int)++h->mail_seq, h->q4_emulate ? h->d_q4_rows.p : (int*)nullptr,
                           h->q4_emulate ? h->d_q4_valid.p : (unsigned char*)nullptr, h->d_entries.p);
    } else {
    // gmask | info live in one allocation; the touch pass (whole GPU), then everything else in one workgroup, which leaves
    // info[] in host-mapped memory and the work area, the launch statistics and the error flags zeroed: no memset, no
    // copy, and the host polls the mapped page for the answer instead of waiting for the stream
    if (h->idxwork_clean_at != h->d_idxwork.p || h->idxwork_clean_n != N) {  // (the build kernel leaves it zeroed for the next round)
        HIPCHK(h, hipMemsetAsync(h->d_idxwork.p, 0, sizeof(int) * ((size_t)N + 8), st));
    }
    h->idxwork_clean_at = nullptr;  // dirty from here until the build kernel has answered
    const long long n_scan = h->n_alert_set >= 0 ? h->n_alert_set : h->n_records_total;
    const dim3 touch_grid((unsigned)std::min<long long>(h->num_cus * 8, (n_scan + 255) / 256));
    // (nothing declared: the delivered boundary records themselves are the alert set -- one more pass over them, which a
    // caller that knows the round's distinct alerts avoids with rapid_sim_set_alert_set; generated streams always declare)
    if (n_scan > 0)
        hipLaunchKernelGGL(rapid::index_touch_kernel, touch_grid, dim3(256), 0, st, h->n_alert_set >= 0 ? h->d_alerts : h->d_records, n_scan, N,
                           (1u << K) - 1u, (long long)h->config_id, h->d_member.p, d_gmask, reinterpret_cast<unsigned int*>(d_info + 4));
    // large populations: the walk over the nodes in several workgroups (two more launches, a tenth of the latency); testing
    // knob bit 12: also for small ones.  Such a round never has direct tables (kIndexChunkedMin nodes do not fit the LDS).
    const int n_chunks = chunked ? (N + rapid::kIndexChunk - 1) / rapid::kIndexChunk : 0;
    if (chunked) {
        HIPCHK(h, h->d_idxblk.ensure((size_t)2 * (size_t)std::max(n_chunks, 1)));
        hipLaunchKernelGGL(rapid::index_count_kernel, dim3((unsigned)n_chunks), dim3(1024), 0, st, d_gmask, N, L, h->d_idxblk.p);
        hipLaunchKernelGGL(rapid::index_assign_kernel, dim3((unsigned)n_chunks), dim3(1024), 0, st, d_gmask, h->d_member.p, N, L,
                           h->d_idxblk.p, h->d_dict.p, h->d_decl.p, h->d_node_of_slot.p, h->d_tbits.p, h->d_trank.p, h->d_tent.p, tent_cap);
        // the hot adjacency slot by slot, one thread each (and the work area cleared by many workgroups instead of one)
        HIPCHK(h, h->d_edges.ensure((size_t)16384 * rapid::kIndexEdgeStride));
        HIPCHK(h, h->d_edge_mask.ensure(16384));
        hipLaunchKernelGGL(rapid::index_edges_kernel, dim3(64), dim3(256), 0, st, h->d_idxblk.p, n_chunks, h->d_node_of_slot.p, h->d_member.p, h->d_obs.p,
                           N, K, h->d_dict.p, h->q4_emulate ? h->d_q4_rows.p : (int*)nullptr, h->q4_emulate ? h->d_q4_valid.p : (unsigned char*)nullptr,
                           h->d_edges.p, h->d_edge_mask.p, d_info, d_gmask);
    }
    hipLaunchKernelGGL(rapid::index_build_block_kernel, dim3(1), dim3(1024), 0, st, d_gmask, h->d_member.p, h->d_obs.p, N, K, L,
                       h->d_dict.p, h->d_decl.p, h->d_node_of_slot.p, h->d_adj_off.p, h->d_adj.p, adj_cap, h->d_tbits.p, h->d_trank.p,
                       h->d_tent.p, tent_cap, d_info, reinterpret_cast<volatile int*>(h->d_mail),
                       ((h->force_exact & (128 | 256 | 8192)) != 0 || chunked) ? -1 : 160 * 1024 - rapid::kBlockStatsBytes,  // (lds_max below)
                       h->d_stats.p, (int)stats_words(h), h->d_errflags.p, (int)++h->mail_seq,
                       chunked ? h->d_idxblk.p : nullptr, n_chunks, h->q4_emulate ? h->d_q4_rows.p : (int*)nullptr,
                       h->q4_emulate ? h->d_q4_valid.p : (unsigned char*)nullptr, chunked ? h->d_edges.p : (const unsigned short*)nullptr,
                       chunked ? h->d_edge_mask.p : (const unsigned short*)nullptr);
    }
    if (timed) HIPCHK(h, hipEventRecord(h->ev_idx1, st));
    HIPCHK(h, hipGetLastError());
    if (int rc = await_mail(h, 15, h->mail_seq)) return rc;
    if (!fused) {
        h->idxwork_clean_at = h->d_idxwork.p;
        h->idxwork_clean_n = N;
    }
    int info[8];
    std::memcpy(info, h->h_mail, sizeof info);  // written by the kernel into host-mapped memory
    h->index_ms_pending = timed;  // the events are read when somebody asks (no wait for them here)
    h->q4_live = (info[2] & 4) != 0;
    if (info[2] & 1) return fail(h, RAPID_ECAPACITY, "the round has %d hot subjects; at most 16318 are supported", info[0]);
    if (info[2] & 2)
        return fail(h, RAPID_ECAPACITY, "hot adjacency has %d entries; at most 65535 are supported", info[3]);
    h->trusted = (info[4] & 1) == 0;
    h->all_down = (info[4] & 2) == 0;
    h->all_current = (info[4] & 4) == 0;
    h->n_slots = info[0];
    h->n_hot = info[1];
    h->n_adj = h->n_hot > 0 ? info[3] : 0;
    h->n_touched = info[5];
    const bool compressed_ok = info[6] != 0;

    // ---- launch geometry: fill the CU's LDS with as many receiver-waves as possible ----
    const int lds_max = 160 * 1024;
    // rounds with thousands of hot subjects keep two slots per LDS word (the detector state decides how many receivers a CU holds)
    h->packed = rapid::tally_wants_packed(h->n_hot) || ((h->force_exact & 8192) != 0 && h->n_hot > 0);  // (testing knob bit 13: packed whatever the size)
    const int per_wave = rapid::tally_wave_bytes(h->n_slots, h->packed);
    const int sh_direct = rapid::tally_shared_bytes(rapid::kDictDirect, N, h->n_touched, h->n_hot, h->n_adj, h->packed);
    const int sh_comp = rapid::tally_shared_bytes(rapid::kDictCompressed, N, h->n_touched, h->n_hot, h->n_adj, h->packed);
    const int sh_mem = rapid::tally_shared_bytes(rapid::kDictMemory, N, h->n_touched, h->n_hot, h->n_adj, h->packed);  // (== kDictResolved: no tables)
    if (sh_mem + per_wave + rapid::kBlockStatsBytes > lds_max)
        return fail(h, RAPID_ECAPACITY, "%d subjects need %d B of LDS per receiver (max %d)", h->n_slots,
                    sh_mem + per_wave + rapid::kBlockStatsBytes, lds_max);
    // Where the node -> slot dictionary lives: in LDS as plain tables (4 B per node) when at least eight receivers still fit
    // next to them; else compressed (3 bits per node + 4 B per node the alert set names: 100,000 nodes in ~25 KB); else in
    // memory.  Testing knob: bit 7 = never direct, bit 8 = never in LDS at all.
    const bool no_direct = (h->force_exact & (128 | 256)) != 0 || info[7] == 0, no_lds = (h->force_exact & 256) != 0;  // info[7]: the build kernel's own verdict
    // Packed rounds over boundary records look their subjects up in memory (one dict_entry per node, gathered through L2).  Knob bit
    // 20: in LDS instead, as hashed buckets of one-byte remainders (kDictHashed) -- when every named subject is hot (a miss is then a
    // report the alert set does not cover, nothing else), the key fits kHashMaxKeyBits and at least three receivers still fit the
    // CU next to it.  Exact and parity-tested, but no faster at 10^6 nodes (1.46 against 1.40 ms per 1.5 x 10^8 records): the exact
    // lookup costs ~50 vector instructions per record where the gather costs the CU's address-coalescing time, and with one wave
    // per SIMD neither hides behind anything (profiles/r05_c5_hashed_dictionary.txt) -- so it is not the default.
    const int sh_hash = rapid::tally_shared_bytes(rapid::kDictHashed, N, h->n_hot, h->n_hot, h->n_adj, true);
    const bool hash_ok = h->packed && h->rec_fmt == rapid::kFmtBoundary && h->n_hot > 0 && h->n_touched == h->n_hot && (h->force_exact & 1048576) != 0 &&
                         rapid::hash_key_bits(N) <= rapid::kHashMaxKeyBits && sh_hash + 3 * per_wave + rapid::kBlockStatsBytes <= lds_max;
    if (h->rec_fmt == rapid::kFmtResident)
        h->dict_mode = rapid::kDictResolved;  // generated records carry their subjects' entries
    else if (hash_ok)
        h->dict_mode = rapid::kDictHashed;
    else if (h->packed)
        h->dict_mode = rapid::kDictMemory;    // (the LDS goes to the receivers' state)
    else if (!no_direct && sh_direct + 8 * per_wave + rapid::kBlockStatsBytes <= lds_max)
        h->dict_mode = rapid::kDictDirect;
    else if (!no_lds && compressed_ok && sh_comp + 8 * per_wave + rapid::kBlockStatsBytes <= lds_max)
        h->dict_mode = rapid::kDictCompressed;
    else
        h->dict_mode = rapid::kDictMemory;
    h->tables_in_lds = h->dict_mode == rapid::kDictDirect || h->dict_mode == rapid::kDictCompressed;
    // (the one-launch index has written them already; compressed tables hold their own entries)
    if (!fused && h->dict_mode != rapid::kDictCompressed) {
        hipLaunchKernelGGL(rapid::dict_entries_kernel, dim3(grid_for((long long)N + 1, 256)), dim3(256), 0, st, h->d_dict.p, h->d_decl.p, N, h->n_hot,
                           h->d_entries.p);
    }
    if (h->dict_mode == rapid::kDictHashed) {
        // one more launch and one more answer to wait for (~20 us on a round of milliseconds): the hot subjects renumbered in the
        // order the hashed lookup yields, the per-slot tables and the triples rewritten into their second buffers
        const int nb = rapid::hash_buckets(N);
        HIPCHK(h, h->d_hoff.ensure((size_t)nb + 4));
        HIPCHK(h, h->d_hrem.ensure(((size_t)h->n_hot + rapid::kHashPad + 3) / 4 + 1));
        HIPCHK(h, h->d_hmem.ensure(((size_t)h->n_hot + 31) / 32 + 1));
        HIPCHK(h, h->d_hnew.ensure((size_t)h->n_hot + 1));
        HIPCHK(h, h->d_nos2.ensure((size_t)N));
        HIPCHK(h, h->d_smask2.ensure((size_t)N + 1 + 8));
        HIPCHK(h, h->d_adj2.ensure((size_t)adj_cap + 1));
        const size_t hl = (size_t)rapid::index_hash_lds_bytes(N, h->n_hot);
        if (!h->hash_lds_attr_set) {
            HIPCHK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(rapid::index_hash_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
            h->hash_lds_attr_set = true;
        }
        bool hashed = false;
        if (hl <= 96 * 1024) {
            hipLaunchKernelGGL(rapid::index_hash_kernel, dim3(1), dim3(1024), hl, st, h->d_node_of_slot.p, h->d_adj_off.p, h->d_adj.p, h->n_hot, h->n_adj, N,
                               h->d_member.p, h->d_hoff.p, reinterpret_cast<unsigned char*>(h->d_hrem.p), h->d_hmem.p, h->d_nos2.p, h->d_smask2.p, h->d_adj2.p,
                               h->d_hnew.p, h->d_entries.p, h->d_dict.p, reinterpret_cast<volatile int*>(h->d_mail), (int)++h->mail_seq);
            HIPCHK(h, hipGetLastError());
            if (int rc = await_mail(h, 11, h->mail_seq)) return rc;
            int ans = 0;
            std::memcpy(&ans, h->h_mail + 48, sizeof ans);  // (word 12: 1 + the multiplier's index, 0: none fits)
            hashed = ans >= 1;
            if (hashed) {
                h->hash_mul = rapid::hash_multiplier(ans - 1);
                std::swap(h->d_node_of_slot, h->d_nos2);
                std::swap(h->d_adj_off, h->d_smask2);
                std::swap(h->d_adj, h->d_adj2);
            }
        }
        if (!hashed) h->dict_mode = rapid::kDictMemory;  // (no multiplier keeps every bucket within its capacity: the dictionary stays in memory)
    }
    const int sh = h->dict_mode == rapid::kDictDirect ? sh_direct : h->dict_mode == rapid::kDictCompressed ? sh_comp : h->dict_mode == rapid::kDictHashed ? sh_hash : sh_mem;
    // Waves per CU (one workgroup per CU, its receivers claimed by its waves from a counter in LDS).  A CU's share of
    // the memory system is saturated by the stream loads of ~7 waves; with w waves a CU works through its n receivers in
    // floor(n / w) full rounds, each as long as w streams sharing the CU's bandwidth, plus a last round of the m
    // remaining receivers that is never shorter than what ~7 streams would take (fewer waves do not stream faster).
    // Among the wave counts that fit, the one with the smallest total wins (ties: more waves) -- e.g. 13 rather than 16
    // waves for 9,492 receivers on 256 CUs (2 x 13 + 11 instead of 2 x 16 + 5 -> 2 x 16 + 7).
    // A population smaller than waves x CUs is spread over as many CUs as it has groups of w receivers: among equal
    // costs the launch with more workgroups wins (1,186 receivers -- C3b's share on one of eight GPUs -- run as 238
    // workgroups of 5 waves, not as 80 of 15), then the one with more waves.
    int best_w = 1;
    long long best_blocks = 0;
    double best_cost = 1e300;
    int w_cap = rapid::tally_max_waves(h->dict_mode, tally_is_trusted(h), h->rec_fmt, h->packed);  // (what decides the instantiation is known by now)
    if (const char* e = env_knob("RAPID_TALLY_WAVES")) w_cap = std::max(1, std::min(w_cap, atoi(e)));  // profiling knob
    const double sat = 7.0;
    for (int w = 1; w <= w_cap; ++w) {
        if (sh + w * per_wave + rapid::kBlockStatsBytes > lds_max) break;
        const long long blocks = std::max<long long>(1, std::min<long long>(((long long)h->n_receivers + w - 1) / w, (long long)h->num_cus));
        const double n_per_cu = (double)h->n_receivers / (double)blocks;
        const double full = std::floor(n_per_cu / w), rem = n_per_cu - full * w;
        const double cost = full * std::max((double)w, sat) + (rem > 0.0 ? std::max(rem, sat) : 0.0);
        if (cost < best_cost - 1e-9 || (cost <= best_cost + 1e-9 && blocks >= best_blocks)) {
            best_cost = cost;
            best_w = w;
            best_blocks = blocks;
        }
    }
    if (const char* e = env_knob("RAPID_TALLY_WAVES_EXACT")) {  // profiling knob: this many waves, whatever the cost model says
        const int w = atoi(e);
        if (w >= 1 && w <= w_cap && sh + w * per_wave + rapid::kBlockStatsBytes <= lds_max) best_w = w;
    }
    h->waves_per_block = best_w;
    h->lds_bytes = sh + best_w * per_wave + rapid::kBlockStatsBytes;
    const long long want = ((long long)h->n_receivers + best_w - 1) / best_w;
    int blocks_per_cu = 1;
    if (const char* e = env_knob("RAPID_TALLY_BLOCKS_PER_CU")) blocks_per_cu = std::max(1, std::min(4, atoi(e)));  // profiling knob
    h->grid_blocks = (int)std::max<long long>(1, std::min<long long>(want, (long long)h->num_cus * blocks_per_cu));
    h->index_valid = true;
    return RAPID_OK;
}

// kTrusted = a delivered record that fails the membership filter or names a ring the index was not built for is an ERROR of
// the stream (RAPID_EINVAL) instead of being dropped per delivery.  It runs only on VERIFIED facts: either every delivered record
// went through the validation pass itself (nothing declared: index_touch_kernel saw them all and none failed), or the declared
// alerts all pass the filter under the current view and the caller vouches that the deliveries are copies of them.  The
// configuration id is compared per delivery by the kernel either way (boundary records; another id: dropped, as
// R/MembershipService.java:653-657 does); generated records rest on the alert set they were generated from.  Testing knob bit 6: never.
bool tally_is_trusted(const rapid_engine* h) {
    if (!h->trusted || (h->force_exact & 64) != 0) return false;
    if (h->n_alert_set < 0) return true;
    if (h->rec_fmt == rapid::kFmtResident) return h->trust_copies && h->gen_clean;
    return h->trust_copies;
}

// Boundary records whose configuration ids need not be read (tally_kernel.h: kCurrent): pre-validated, every alert of the round
// index of the engine's configuration, and every delivered record KNOWN to be a copy of one of them -- because the library laid the
// records down itself (rapid_sim_generate / rapid_sim_round_tiled), or because the caller says so at level 2 of
// rapid_sim_trust_alert_copies (the one thing on this path that rests on the caller's word: level 1 compares the ids per delivery
// and drops late deliveries of an earlier configuration, R/MembershipService.java:653-657).  Testing knob bit 22: never.
bool records_known_current(const rapid_engine* h) {
    return tally_is_trusted(h) && h->all_current && h->n_alert_set >= 0 && h->rec_fmt == rapid::kFmtBoundary && (h->force_exact & 4194304) == 0 &&
           (h->streams_generated || h->no_late_copies);
}

int launch_tally(rapid_engine* h) {
    rapid::TallyParams p;
    p.core = h->d_records;
    p.rec_off = h->d_rec_off;
    p.n_receivers = h->n_receivers;
    p.n_nodes = h->n_nodes;
    p.K = h->cfg.K;
    p.H = h->cfg.H;
    p.L = h->cfg.L;
    p.cfg_id = h->config_id;
    p.idx.dict = h->d_dict.p;
    p.idx.decl = h->d_decl.p;
    p.idx.tbits = h->d_tbits.p;
    p.idx.trank = h->d_trank.p;
    p.idx.tent = h->d_tent.p;
    p.idx.n_touched = h->n_touched;
    p.idx.entries = h->d_entries.p;
    p.idx.hoff = h->d_hoff.p;
    p.idx.hrem = reinterpret_cast<const unsigned char*>(h->d_hrem.p);
    p.idx.hmem = h->d_hmem.p;
    p.idx.hmul = h->hash_mul;
    p.idx.hbits = rapid::hash_key_bits(h->n_nodes);
    p.error_flags = h->d_errflags.p;
    p.stream_bytes = h->records_bytes;
    p.idx.node_of_slot = h->d_node_of_slot.p;
    p.idx.smask = h->d_adj_off.p;  // (the buffers keep their round-1 names: per-slot masks, flat triple list)
    p.idx.pairs = h->d_adj.p;
    p.idx.n_hot = h->n_hot;
    p.idx.n_adj = h->n_adj;
    p.emit_batch = h->d_emit.p + h->out_base;  // (a tiled round: the running tile's place in the round's per-receiver results)
    p.num_proposals = h->d_nprop.p + h->out_base;
    p.prop_count = h->d_pcount.p + h->out_base;
    p.fingerprint = h->d_fp.p + h->out_base;
    p.props = h->d_props.p;
    p.prop_cap = h->max_cut;
    p.stats = h->d_stats.p;  // [grid_blocks][8]
    p.waves_per_block = h->waves_per_block;
    p.flags = h->force_exact & (1 | 4 | 8 | 32);
    if (const char* e = env_knob("RAPID_SETTLE_SKIP")) p.flags |= (atoi(e) & 7) << 10;  // measurement knob (test build): parts of the in-launch vote settlement left out
    p.stagger = 0;
    if (const char* e = env_knob("RAPID_TALLY_STAGGER")) p.stagger = std::max(0, std::min(64, atoi(e)));  // profiling knob
    // The last eighth of the receivers is not dealt to the workgroups but left in a common pool (tally_kernel.h: n_static),
    // once a population is at least two rounds of the launch; testing knob bit 10: everything dealt statically.
    p.n_static = h->n_receivers;
    p.pool = nullptr;
    // the words behind the launch's statistics rows: [0] = pool, [1..5] = vote accumulators (zeroed with the rows)
    p.vote_acc = h->d_stats.p + (size_t)8 * (size_t)h->grid_blocks + 1;
    p.vote_res = h->d_voteback.p;
    h->tally_votes_valid = true;
    // the voters' proposals as bitmaps over the round's hot slots, for the verification that follows (grown when a round needs more)
    p.bitmap_words = (h->n_hot + 63) / 64;
    HIPCHK(h, h->d_bitmaps.ensure((size_t)std::max(h->n_receivers, 1) * (size_t)std::max(p.bitmap_words, 1)));
    p.bitmaps = h->d_bitmaps.p;
    h->tally_bitmaps_valid = true;
    h->bitmap_words = p.bitmap_words;
    // The fast round settled by the launch itself -- OPT-IN (knob bit 23 of rapid_sim_set_force_exact).  Measured at C3b on two boxes
    // (profiles/r06_ab_settle_in_tally.txt): the verification launch (12 us) disappears from the round's path, the tally launch grows
    // by 6-11 us (its last workgroup's chain of round trips: candidate, comparison, totals, answer), the round gets 3-5 us shorter
    // (1 %) and the dominant kernel 2-3 % longer -- not a trade the default should make.
    p.vote_cand = nullptr;
    p.vote_deferred = nullptr;
    p.vote_deferred_cap = 0;
    p.vote_publish = nullptr;
    p.vote_seq_out = nullptr;
    p.vote_seq = 0u;
    h->tally_settled_valid = false;
    if (h->settle_in_tally && h->out_base == 0 && p.bitmap_words <= 64 && h->n_receivers <= 262144 && (h->force_exact & 8388608) != 0 &&
        (h->force_exact & (2048 | 262144 | 512)) == 0) {
        constexpr int kVoteDeferredCap = 4096;
        if (!h->d_vote_cand.p) {
            HIPCHK(h, h->d_vote_cand.ensure(4 + 64));
            HIPCHK(h, h->d_vote_deferred.ensure(1 + kVoteDeferredCap));
            HIPCHK(h, hipMemsetAsync(h->d_vote_cand.p, 0, (4 + 64) * 8, h->stream));  // (once: every launch leaves them zeroed)
            HIPCHK(h, hipMemsetAsync(h->d_vote_deferred.p, 0, (1 + kVoteDeferredCap) * 4, h->stream));
        }
        p.vote_cand = h->d_vote_cand.p;
        p.vote_deferred = h->d_vote_deferred.p;
        p.vote_deferred_cap = kVoteDeferredCap;
        if (!h->comm) {  // one rank holds the population: the answer goes straight to the page the host polls
            if (int rc = ensure_mailbox(h)) return rc;
            p.vote_publish = reinterpret_cast<volatile unsigned long long*>(h->d_mail + 64);
            p.vote_seq_out = reinterpret_cast<volatile unsigned int*>(h->d_mail) + 14;
            p.vote_seq = ++h->mail_seq;
            h->tally_settled_seq = p.vote_seq;
        }
        h->tally_settled_valid = true;
    }
    {
        const long long slots = (long long)h->grid_blocks * h->waves_per_block;
        const char* e = env_knob("RAPID_POOL_EIGHTHS");  // measurement knob: size of the pool in eighths of the population
        const int eighths = e ? std::max(0, std::min(7, atoi(e))) : 1;
        if ((h->force_exact & 1024) == 0 && eighths > 0 && (long long)h->n_receivers >= 2 * slots) {
            long long ns = ((long long)h->n_receivers * (8 - eighths) / 8 / h->grid_blocks) * h->grid_blocks;
            ns = std::max(ns, slots);
            p.n_static = (int)ns;
            p.pool = reinterpret_cast<unsigned int*>(h->d_stats.p + (size_t)8 * (size_t)h->grid_blocks);
        }
    }
    const dim3 grid((unsigned)h->grid_blocks), block((unsigned)h->waves_per_block * 64u);
    // pre-validated instantiation: the scanned alerts all pass the filter AND (when they are a declared set rather than the
    // delivered records themselves) the caller vouches that the deliveries are copies of them; bit6 of the testing knob: never
    const bool trusted = tally_is_trusted(h);
    const size_t lds = (size_t)h->lds_bytes;
    using namespace rapid;
    // pre-validated boundary records of ONE configuration -- the engine's: their configuration ids stay in the cache lines
    // (tally_kernel.h: kCurrent; testing knob bit 22: compared per delivery all the same)
    if (records_known_current(h)) {
        if (h->packed && h->dict_mode == kDictMemory) {
            hipLaunchKernelGGL((tally_population_kernel<kDictMemory, true, kFmtBoundary, true, true>), grid, block, lds, h->stream, p);
            return RAPID_OK;
        }
        if (!h->packed && h->dict_mode == kDictMemory) {
            hipLaunchKernelGGL((tally_population_kernel<kDictMemory, true, kFmtBoundary, false, true>), grid, block, lds, h->stream, p);
            return RAPID_OK;
        }
        if (!h->packed && h->dict_mode == kDictDirect) {
            hipLaunchKernelGGL((tally_population_kernel<kDictDirect, true, kFmtBoundary, false, true>), grid, block, lds, h->stream, p);
            return RAPID_OK;
        }
        if (!h->packed && h->dict_mode == kDictCompressed) {
            hipLaunchKernelGGL((tally_population_kernel<kDictCompressed, true, kFmtBoundary, false, true>), grid, block, lds, h->stream, p);
            return RAPID_OK;
        }
    }
    switch ((h->packed ? 16 : 0) + (h->rec_fmt == kFmtBoundary ? 0 : 8) + h->dict_mode * 2 + (trusted ? 1 : 0)) {
        case 24: hipLaunchKernelGGL((tally_population_kernel<kDictHashed, false, kFmtBoundary, true>), grid, block, lds, h->stream, p); break;
        case 25: hipLaunchKernelGGL((tally_population_kernel<kDictHashed, true, kFmtBoundary, true>), grid, block, lds, h->stream, p); break;
        case 16: hipLaunchKernelGGL((tally_population_kernel<kDictMemory, false, kFmtBoundary, true>), grid, block, lds, h->stream, p); break;
        case 17: hipLaunchKernelGGL((tally_population_kernel<kDictMemory, true, kFmtBoundary, true>), grid, block, lds, h->stream, p); break;
        case 30: hipLaunchKernelGGL((tally_population_kernel<kDictResolved, false, kFmtResident, true>), grid, block, lds, h->stream, p); break;
        case 31: hipLaunchKernelGGL((tally_population_kernel<kDictResolved, true, kFmtResident, true>), grid, block, lds, h->stream, p); break;
        case 0: hipLaunchKernelGGL((tally_population_kernel<kDictMemory, false, kFmtBoundary>), grid, block, lds, h->stream, p); break;
        case 1: hipLaunchKernelGGL((tally_population_kernel<kDictMemory, true, kFmtBoundary>), grid, block, lds, h->stream, p); break;
        case 2: hipLaunchKernelGGL((tally_population_kernel<kDictDirect, false, kFmtBoundary>), grid, block, lds, h->stream, p); break;
        case 3: hipLaunchKernelGGL((tally_population_kernel<kDictDirect, true, kFmtBoundary>), grid, block, lds, h->stream, p); break;
        case 4: hipLaunchKernelGGL((tally_population_kernel<kDictCompressed, false, kFmtBoundary>), grid, block, lds, h->stream, p); break;
        case 5: hipLaunchKernelGGL((tally_population_kernel<kDictCompressed, true, kFmtBoundary>), grid, block, lds, h->stream, p); break;
        case 14: hipLaunchKernelGGL((tally_population_kernel<kDictResolved, false, kFmtResident>), grid, block, lds, h->stream, p); break;
        case 15: hipLaunchKernelGGL((tally_population_kernel<kDictResolved, true, kFmtResident>), grid, block, lds, h->stream, p); break;
        default: return fail(h, RAPID_ESTATE, "no tally kernel for record format %d, dictionary mode %d", h->rec_fmt, h->dict_mode);
    }
    return RAPID_OK;
}

int prepare_tally(rapid_engine* h) {
    if (!h->view_built) return fail(h, RAPID_ESTATE, "view not built");
    if (!h->streams_loaded) return fail(h, RAPID_ESTATE, "no alert streams loaded");
    HIPCHK(h, h->d_errflags.ensure(2));
    HIPCHK(h, h->d_stats.ensure(stats_words(h)));
    HIPCHK(h, h->d_voteback.ensure((10 * 8 + ((size_t)h->max_cut + 1) * sizeof(int) + 7) / 8));  // launch_tally: vote_res
    if (h->rec_fmt == rapid::kFmtResident && h->gen_cfg_id != h->config_id)
        return fail(h, RAPID_ESTATE, "the view changed since these deliveries were generated: call rapid_sim_generate again");
    h->stats_fresh = false;
    if (!h->index_valid) {
        int rc = build_round_index(h);  // also zeroes the error flags and the launch statistics / pool words
        if (rc) return rc;
        h->stats_fresh = true;
    }
    if (int rc = ensure_tally_attrs(h)) return rc;
    const size_t R = (size_t)std::max(h->n_receivers, 1);
    HIPCHK(h, h->d_emit.ensure(R));
    HIPCHK(h, h->d_nprop.ensure(R));
    HIPCHK(h, h->d_pcount.ensure(R));
    HIPCHK(h, h->d_fp.ensure(R));
    HIPCHK(h, h->d_props.ensure(R * (size_t)h->max_cut));
    HIPCHK(h, h->d_next.ensure(4));
    return RAPID_OK;
}

int ensure_tally_attrs(rapid_engine* h) {
    if (!h->lds_attr_set) {  // once per engine: every instantiation may use the whole 160 KiB of LDS
        using namespace rapid;
        const void* kernels[18] = {reinterpret_cast<const void*>(tally_population_kernel<kDictMemory, true, kFmtBoundary, true, true>),
                                  reinterpret_cast<const void*>(tally_population_kernel<kDictMemory, true, kFmtBoundary, false, true>),
                                  reinterpret_cast<const void*>(tally_population_kernel<kDictDirect, true, kFmtBoundary, false, true>),
                                  reinterpret_cast<const void*>(tally_population_kernel<kDictCompressed, true, kFmtBoundary, false, true>),
                                  reinterpret_cast<const void*>(tally_population_kernel<kDictHashed, false, kFmtBoundary, true>),
                                  reinterpret_cast<const void*>(tally_population_kernel<kDictHashed, true, kFmtBoundary, true>),
                                  reinterpret_cast<const void*>(tally_population_kernel<kDictResolved, false, kFmtResident, true>),
                                  reinterpret_cast<const void*>(tally_population_kernel<kDictResolved, true, kFmtResident, true>),
                                  reinterpret_cast<const void*>(tally_population_kernel<kDictMemory, false, kFmtBoundary, true>),
                                  reinterpret_cast<const void*>(tally_population_kernel<kDictMemory, true, kFmtBoundary, true>),
                                  reinterpret_cast<const void*>(tally_population_kernel<kDictResolved, false, kFmtResident>),
                                  reinterpret_cast<const void*>(tally_population_kernel<kDictResolved, true, kFmtResident>),
                                  reinterpret_cast<const void*>(tally_population_kernel<kDictMemory, false, kFmtBoundary>),
                                  reinterpret_cast<const void*>(tally_population_kernel<kDictMemory, true, kFmtBoundary>),
                                  reinterpret_cast<const void*>(tally_population_kernel<kDictDirect, false, kFmtBoundary>),
                                  reinterpret_cast<const void*>(tally_population_kernel<kDictDirect, true, kFmtBoundary>),
                                  reinterpret_cast<const void*>(tally_population_kernel<kDictCompressed, false, kFmtBoundary>),
                                  reinterpret_cast<const void*>(tally_population_kernel<kDictCompressed, true, kFmtBoundary>)};
        for (const void* k : kernels) HIPCHK(h, hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        h->lds_attr_set = true;
    }
    return RAPID_OK;
}

}  // namespace

extern "C" {

int rapid_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    int usable = 0;
    for (int i = 0; i < n; ++i) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, i) == hipSuccess && std::strncmp(prop.gcnArchName, "gfx950", 6) == 0) ++usable;
    }
    return usable;
}

int rapid_engine_create(const rapid_engine_config* cfg, rapid_engine** out) {
    if (!cfg || !out) return RAPID_EINVAL;
    *out = nullptr;
    if (!valid_khl(cfg->K, cfg->H, cfg->L) || cfg->K > RAPID_MAX_K || cfg->n_max <= 0) return RAPID_EINVAL;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || cfg->device_id < 0 || cfg->device_id >= n) return RAPID_EDEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, cfg->device_id) != hipSuccess) return RAPID_EDEVICE;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return RAPID_EDEVICE;  // kernels exist for gfx950 only
    if (hipSetDevice(cfg->device_id) != hipSuccess) return RAPID_EDEVICE;
    rapid_engine* h = new rapid_engine();
    h->cfg = *cfg;
    h->max_cut = cfg->max_cut > 0 ? cfg->max_cut : std::min(cfg->n_max, 4096);
    h->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
        delete h;
        return RAPID_EDEVICE;
    }
    if (ensure_arena(h) != RAPID_OK) {
        (void)hipStreamDestroy(h->stream);
        (void)hipGetLastError();
        delete h;
        return RAPID_EDEVICE;
    }
    *out = h;
    return RAPID_OK;
}

void rapid_engine_destroy(rapid_engine* h) {
    if (!h) return;
    (void)hipSetDevice(h->cfg.device_id);
    // a failure here has nobody to be reported to, but it must not be left behind as the thread's "last error" for an
    // unrelated later call (rocPRIM checks hipGetLastError after its launches) to trip over
    auto quiet = [](hipError_t e, const char* what) {
        if (e != hipSuccess) {
            if (env_knob("RAPID_DEBUG")) fprintf(stderr, "rapid_engine_destroy: %s: %s\n", what, hipGetErrorString(e));
            (void)hipGetLastError();
        }
    };
    if (h->comm) (void)ncclCommDestroy(h->comm);
    if (h->stream) quiet(hipStreamSynchronize(h->stream), "hipStreamSynchronize");
    if (h->ev_idx0) quiet(hipEventDestroy(h->ev_idx0), "hipEventDestroy");
    if (h->ev_idx1) quiet(hipEventDestroy(h->ev_idx1), "hipEventDestroy");
    if (h->ev0) quiet(hipEventDestroy(h->ev0), "hipEventDestroy");
    if (h->ev1) quiet(hipEventDestroy(h->ev1), "hipEventDestroy");
    if (h->h_arena) quiet(hipHostFree(h->h_arena), "hipHostFree(mailbox + view staging)");
    if (h->h_alert_stage) quiet(hipHostFree(h->h_alert_stage), "hipHostFree(alert staging)");
    if (h->ev_alert) quiet(hipEventDestroy(h->ev_alert), "hipEventDestroy");
    for (hipEvent_t e : {h->ev_gen_ready[0], h->ev_gen_ready[1], h->ev_gen_free[0], h->ev_gen_free[1], h->ev_gen_inputs})
        if (e) quiet(hipEventDestroy(e), "hipEventDestroy");
    if (h->stream_gen) quiet(hipStreamDestroy(h->stream_gen), "hipStreamDestroy");
    if (h->stream) quiet(hipStreamDestroy(h->stream), "hipStreamDestroy");
    h->d_blob.release(); h->d_host_off.release(); h->d_ports.release(); h->d_keys.release();
    h->d_hx_host0.release(); h->d_hx_port0.release(); h->d_member.release(); h->d_members.release();
    h->d_sort_keys.release(); h->d_sort_vals.release(); h->d_ring_skeys.release(); h->d_ring.release();
    h->d_pos.release(); h->d_obs.release(); h->d_subj.release();
    h->d_ids_hi.release(); h->d_ids_lo.release(); h->d_cfg_out.release(); h->d_sort_tmp.release();
    h->d_q4_nodes.release(); h->d_q4_rows.release(); h->d_q4_valid.release(); h->d_q4_flag.release(); h->d_entries.release();
    h->d_gen_res.release(); h->d_gen_keep.release(); h->d_gen_boff.release(); h->d_gen_rx.release(); h->d_gen_bat.release();
    h->d_ids_hi2.release(); h->d_ids_lo2.release(); h->d_ids_new.release(); h->d_cfg_partial.release(); h->d_chunk_kept.release();
    h->d_bitmaps.release();
    h->d_vote_cand.release(); h->d_vote_deferred.release();
    h->d_vacc.release();
    h->d_idacc.release(); h->d_chunk_base.release(); h->d_chunk_lb.release(); h->d_nonmembers.release();
    h->d_hoff.release(); h->d_hnew.release(); h->d_smask2.release(); h->d_hrem.release(); h->d_hmem.release(); h->d_adj2.release(); h->d_nos2.release();
    h->d_gone.release();
    h->d_edges.release(); h->d_edge_mask.release();
    h->d_joiners.release(); h->d_join_nodes.release(); h->d_join_vals.release(); h->d_join_keys.release(); h->d_join_skeys.release();
    h->d_records_own.release(); h->d_rec_off_own.release(); h->d_emit.release(); h->d_nprop.release();
    h->d_pcount.release(); h->d_props.release(); h->d_fp.release(); h->d_stats.release();
    h->d_alert_set.release(); h->d_next.release(); h->d_idxwork.release(); h->d_idxblk.release(); h->d_adj.release(); h->d_dict.release(); h->d_decl.release(); h->d_errflags.release(); h->d_trank.release(); h->d_tbits.release(); h->d_tent.release();
    h->d_adj_off.release(); h->d_node_of_slot.release(); h->d_loadflags.release();
    h->d_hist.release(); h->d_winner.release(); h->d_mm.release(); h->d_mismatch.release(); h->d_ref.release(); h->d_voteback.release(); h->d_gather.release();
    (void)hipGetLastError();
    delete h;
}

const char* rapid_last_error(const rapid_engine* h) { return h ? h->err.c_str() : "null engine"; }

// A known-answer test of the whole path on h's device, through the public entry points, on a PRIVATE engine (h's view, streams and
// results are not touched): five endpoints 10.0.0.0 .. 10.0.0.4 : 5000 with NodeIds (i + 1, i + 101), K = H = 3, L = 1 -- the
// view's configuration id and ring 0 must be the constants below (the CPU oracle's, restated by tests/test_abi.py so that they cannot
// drift); node 4 is then reported DOWN on every ring to the four other receivers, the fast round must decide the cut {4} with four of
// four votes, and the configuration id after the cut must be the second constant.  RAPID_OK, or RAPID_EDEVICE with the first
// difference in rapid_last_error(h).  A host calls it once after rapid_engine_create: a runtime or a device that cannot do this is
// found out by an error code, before the first real view is built (INTEGRATION.md section 6 -- a device FAULT still ends the process;
// nothing in a process can catch that).
int rapid_engine_self_test(rapid_engine* h) {
    if (!h) return RAPID_EINVAL;
    static const int64_t kCfg = 886890972789580738ll, kCfgAfter = -3371242312397245251ll;
    static const int32_t kRing0[5] = {4, 1, 2, 0, 3}, kRing0After[4] = {1, 2, 0, 3};
    rapid_engine_config cfg = h->cfg;
    cfg.n_max = 8;
    cfg.K = 3;
    cfg.H = 3;
    cfg.L = 1;
    cfg.max_cut = 0;
    rapid_engine* t = nullptr;
    int rc = rapid_engine_create(&cfg, &t);
    if (rc != RAPID_OK || !t) return fail(h, RAPID_EDEVICE, "self test: a second engine could not be created on device %d (%d)", h->cfg.device_id, rc);
    std::string why;
    auto bad = [&](const char* fmt, ...) {
        char buf[400];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        why = buf;
        return RAPID_EDEVICE;
    };
    auto run = [&]() -> int {
        uint8_t blob[5 * 8];
        int32_t off[6] = {0, 0, 0, 0, 0, 0}, ports[5], members[5];
        int64_t hi[5], lo[5];
        for (int i = 0; i < 5; ++i) {
            const int len = snprintf(reinterpret_cast<char*>(blob) + off[i], 9, "10.0.0.%d", i);
            off[i + 1] = off[i] + len;
            ports[i] = 5000;
            members[i] = i;
            hi[i] = i + 1;
            lo[i] = i + 101;
        }
        int r = rapid_view_build(t, blob, off, ports, hi, lo, 5, members, 5, nullptr, nullptr, 0);
        if (r) return bad("rapid_view_build: %d (%s)", r, rapid_last_error(t));
        int64_t id = 0;
        if ((r = rapid_view_config_id(t, &id))) return bad("rapid_view_config_id: %d (%s)", r, rapid_last_error(t));
        if (id != kCfg) return bad("configuration id of the 5-node view is %lld, expected %lld", (long long)id, (long long)kCfg);
        int32_t ring[8], n = 0;
        if ((r = rapid_view_ring(t, 0, ring, 8, &n))) return bad("rapid_view_ring: %d (%s)", r, rapid_last_error(t));
        if (n != 5 || std::memcmp(ring, kRing0, sizeof kRing0) != 0) return bad("ring 0 of the 5-node view differs from the known answer");
        int32_t obs[4];
        if ((r = rapid_view_observers(t, 4, obs, 4, &n)) || n != 3) return bad("rapid_view_observers(4): %d, %d observers (%s)", r, n, rapid_last_error(t));
        rapid_alert_record recs[4 * 3];
        int64_t rec_off[5];
        for (int rx = 0; rx < 4; ++rx) {
            rec_off[rx] = 3 * rx;
            for (int k = 0; k < 3; ++k) {
                rapid_alert_record& a = recs[3 * rx + k];
                std::memset(&a, 0, sizeof a);
                a.cfg_id = kCfg;
                a.src = (uint32_t)obs[k];
                a.dst = 4u;
                a.ring_mask = (uint16_t)(1u << k);
                a.status = RAPID_EDGE_DOWN;
                a.flags = k == 2 ? RAPID_ALERT_LAST_IN_BATCH : 0u;
            }
        }
        rec_off[4] = 12;
        if ((r = rapid_sim_load_streams(t, recs, rec_off, 4))) return bad("rapid_sim_load_streams: %d (%s)", r, rapid_last_error(t));
        rapid_round_result rr;
        std::memset(&rr, 0, sizeof rr);
        int64_t after = 0;
        if ((r = rapid_sim_round(t, 1, &rr, &after))) return bad("rapid_sim_round: %d (%s)", r, rapid_last_error(t));
        if (rr.decided != 1 || rr.cut_size != 1 || rr.votes_winner != 4 || rr.quorum != 4)
            return bad("the round decided %d, cut of %d, %lld votes, quorum %d; expected a decided cut of 1 with 4 of 4", rr.decided, rr.cut_size,
                       (long long)rr.votes_winner, rr.quorum);
        int32_t cut[4];
        if ((r = rapid_sim_decided_cut(t, cut, 4, &n)) || n != 1 || cut[0] != 4) return bad("decided cut is not {4} (%d, n=%d)", r, n);
        if (after != kCfgAfter) return bad("configuration id after the cut is %lld, expected %lld", (long long)after, (long long)kCfgAfter);
        if ((r = rapid_view_ring(t, 0, ring, 8, &n)) || n != 4 || std::memcmp(ring, kRing0After, sizeof kRing0After) != 0)
            return bad("ring 0 after the cut differs from the known answer");
        return RAPID_OK;
    };
    rc = run();
    rapid_engine_destroy(t);
    (void)hipSetDevice(h->cfg.device_id);
    if (rc != RAPID_OK) return fail(h, RAPID_EDEVICE, "self test failed: %s", why.c_str());
    return RAPID_OK;
}

// ------------------------------------------------------------------------------------------------ view
int rapid_view_build(rapid_engine* h, const uint8_t* hostnames, const int32_t* host_off, const int32_t* ports,
                     const int64_t* id_hi, const int64_t* id_lo, int32_t n_nodes, const int32_t* members,
                     int32_t n_members, const int64_t* extra_id_hi, const int64_t* extra_id_lo, int32_t n_extra) {
    if (!h) return RAPID_EINVAL;
    if (!hostnames || !host_off || !ports || !id_hi || !id_lo || n_nodes <= 0 || n_nodes > h->cfg.n_max || n_members < 0 ||
        (n_members > 0 && !members) || n_extra < 0 || (n_extra > 0 && (!extra_id_hi || !extra_id_lo)) || host_off[0] < 0)
        return fail(h, RAPID_EINVAL, "bad arguments to rapid_view_build (n_nodes=%d, n_max=%d)", n_nodes, h->cfg.n_max);
    // everything that can be wrong with the arguments is found before the engine's state is touched: a call that returns
    // RAPID_EINVAL leaves a previously built view as it was
    for (int i = 0; i < n_members; ++i)
        if (members[i] < 0 || members[i] >= n_nodes) return fail(h, RAPID_EINVAL, "member index %d out of range", members[i]);
    for (int i = 0; i < n_nodes; ++i)
        if (host_off[i + 1] < host_off[i]) return fail(h, RAPID_EINVAL, "hostname offsets must not decrease (entry %d)", i);
    int rc = use_device(h);
    if (rc) return rc;
    const int K = h->cfg.K;
    if ((rc = presize_view(h))) return rc;
    h->view_built = false;  // (until the new view stands: a device error below leaves "no view", not a mixture of two)
    h->n_nodes = n_nodes;
    h->id_hi.assign(id_hi, id_hi + n_nodes);
    h->id_lo.assign(id_lo, id_lo + n_nodes);
    h->member.assign((size_t)n_nodes, 0);
    std::vector<std::pair<int64_t, int64_t>> ids;
    ids.reserve((size_t)n_members + (size_t)n_extra);
    for (int i = 0; i < n_members; ++i) {
        const int m = members[i];
        h->member[(size_t)m] = 1;  // Set semantics, like TreeSet.addAll (R/MembershipView.java:82-85)
        ids.push_back({id_hi[m], id_lo[m]});
    }
    for (int i = 0; i < n_extra; ++i) ids.push_back({extra_id_hi[i], extra_id_lo[i]});
    std::sort(ids.begin(), ids.end());  // the order of the reference's TreeSet<NodeId> (:474-500); a set: duplicates collapse
    ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
    {
        const size_t ni = ids.size();
        std::vector<long long> hi(ni), lo(ni);
        for (size_t i = 0; i < ni; ++i) {
            hi[i] = ids[i].first;
            lo[i] = ids[i].second;
        }
        HIPCHK(h, h->d_ids_hi.ensure(std::max<size_t>(ni, 1)));
        HIPCHK(h, h->d_ids_lo.ensure(std::max<size_t>(ni, 1)));
        if (ni) {
            HIPCHK(h, hipMemcpyAsync(h->d_ids_hi.p, hi.data(), ni * 8, hipMemcpyHostToDevice, h->stream));
            HIPCHK(h, hipMemcpyAsync(h->d_ids_lo.p, lo.data(), ni * 8, hipMemcpyHostToDevice, h->stream));
            HIPCHK(h, hipStreamSynchronize(h->stream));
        }
        h->n_ids_dev = (int)ni;
    }
    h->ids_pending.clear();
    HIPCHK(h, hipMemsetAsync(h->d_q4_valid.p, 0, (size_t)h->cfg.n_max, h->stream));  // a new MembershipView object: nothing memoised

    const size_t blob_bytes = (size_t)host_off[n_nodes];
    h->reg_blob.assign(hostnames, hostnames + blob_bytes);
    h->reg_off.assign(host_off, host_off + n_nodes + 1);
    h->reg_ports.assign(ports, ports + n_nodes);
    HIPCHK(h, h->d_blob.ensure(std::max<size_t>(blob_bytes, 1)));
    HIPCHK(h, h->d_host_off.ensure((size_t)n_nodes + 1));
    HIPCHK(h, h->d_ports.ensure((size_t)n_nodes));
    HIPCHK(h, h->d_keys.ensure((size_t)K * n_nodes));
    HIPCHK(h, h->d_hx_host0.ensure((size_t)n_nodes));
    HIPCHK(h, h->d_hx_port0.ensure((size_t)n_nodes));
    HIPCHK(h, hipMemcpyAsync(h->d_blob.p, hostnames, blob_bytes, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_host_off.p, host_off, sizeof(int) * ((size_t)n_nodes + 1), hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_ports.p, ports, sizeof(int) * (size_t)n_nodes, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(rapid::ring_keys_kernel, dim3(grid_for((long long)K * n_nodes, 256)), dim3(256), 0, h->stream,
                       h->d_blob.p, h->d_host_off.p, h->d_ports.p, n_nodes, K, h->d_keys.p, h->d_hx_host0.p, h->d_hx_port0.p);
    LAUNCHCHK(h, "ring_keys_kernel");
    h->host_keys0_valid = false;
    HIPCHK(h, hipStreamSynchronize(h->stream));  // borrowed inputs may go away after the call
    h->streams_loaded = false;
    h->ring_member.clear();  // new endpoints, new ring keys: nothing to compact from
    h->changed.clear();
    h->changed_valid = false;
    h->ring_m = 0;
    if ((rc = rebuild_view(h))) return rc;
    h->view_built = true;
    return RAPID_OK;
}

int rapid_view_register_endpoints(rapid_engine* h, const uint8_t* hostnames, const int32_t* host_off, const int32_t* ports,
                                  const int64_t* id_hi, const int64_t* id_lo, int32_t n_new, int32_t* first_index_out) {
    if (!h) return RAPID_EINVAL;
    if (!h->view_built) return fail(h, RAPID_ESTATE, "view not built");
    if (n_new < 0 || (n_new > 0 && (!hostnames || !host_off || !ports || !id_hi || !id_lo)) || (n_new > 0 && host_off[0] < 0))
        return fail(h, RAPID_EINVAL, "bad arguments to rapid_view_register_endpoints");
    if (n_new > h->cfg.n_max - h->n_nodes)
        return fail(h, RAPID_ECAPACITY, "%d registered endpoints + %d new exceed n_max=%d", h->n_nodes, n_new, h->cfg.n_max);
    for (int i = 0; i < n_new; ++i)
        if (host_off[i + 1] < host_off[i]) return fail(h, RAPID_EINVAL, "hostname offsets must not decrease");
    if (first_index_out) *first_index_out = h->n_nodes;
    if (n_new == 0) return RAPID_OK;
    int rc = use_device(h);
    if (rc) return rc;
    const int K = h->cfg.K, n_old = h->n_nodes, n_nodes = n_old + n_new;
    const int base = h->reg_off[(size_t)n_old];
    // Device memory first (the one step that fails for an ordinary reason -- no memory left), the host-side registry after it:
    // a call that returns an error has registered nothing.
    const size_t blob_bytes = h->reg_blob.size() + (size_t)(host_off[n_new] - host_off[0]);
    HIPCHK(h, h->d_blob.ensure(std::max<size_t>(blob_bytes, 1)));
    HIPCHK(h, h->d_host_off.ensure((size_t)n_nodes + 1));
    HIPCHK(h, h->d_ports.ensure((size_t)n_nodes));
    HIPCHK(h, h->d_keys.ensure((size_t)K * n_nodes));
    HIPCHK(h, h->d_hx_host0.ensure((size_t)n_nodes));
    HIPCHK(h, h->d_hx_port0.ensure((size_t)n_nodes));
    h->reg_blob.insert(h->reg_blob.end(), hostnames + host_off[0], hostnames + host_off[n_new]);
    for (int i = 1; i <= n_new; ++i) h->reg_off.push_back(base + (host_off[i] - host_off[0]));
    h->reg_ports.insert(h->reg_ports.end(), ports, ports + n_new);
    h->id_hi.insert(h->id_hi.end(), id_hi, id_hi + n_new);
    h->id_lo.insert(h->id_lo.end(), id_lo, id_lo + n_new);
    h->member.resize((size_t)n_nodes, 0);  // registered, not members: rapid_view_ring_add admits them
    h->n_nodes = n_nodes;
    // the ring keys are laid out [K][n_nodes]: with another stride they are computed again for everybody (a few us per
    // ten thousand endpoints), and the rings are sorted afresh -- the members and their order are the same
    auto upload = [&]() -> int {
        HIPCHK(h, hipMemcpyAsync(h->d_blob.p, h->reg_blob.data(), blob_bytes, hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(h->d_host_off.p, h->reg_off.data(), sizeof(int) * ((size_t)n_nodes + 1), hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipMemcpyAsync(h->d_ports.p, h->reg_ports.data(), sizeof(int) * (size_t)n_nodes, hipMemcpyHostToDevice, h->stream));
        hipLaunchKernelGGL(rapid::ring_keys_kernel, dim3(grid_for((long long)K * n_nodes, 256)), dim3(256), 0, h->stream,
                           h->d_blob.p, h->d_host_off.p, h->d_ports.p, n_nodes, K, h->d_keys.p, h->d_hx_host0.p, h->d_hx_port0.p);
        h->host_keys0_valid = false;
        HIPCHK(h, hipStreamSynchronize(h->stream));
        return RAPID_OK;
    };
    if ((rc = upload())) {  // (a device fault: the registry goes back to where it was; the key table may be half rewritten, so the view has to be built again)
        h->reg_blob.resize((size_t)base);
        h->reg_off.resize((size_t)n_old + 1);
        h->reg_ports.resize((size_t)n_old);
        h->id_hi.resize((size_t)n_old);
        h->id_lo.resize((size_t)n_old);
        h->member.resize((size_t)n_old);
        h->n_nodes = n_old;
        h->view_built = false;
        return rc;
    }
    h->host_tables_valid = false;
    h->ring_member.clear();
    h->changed.clear();
    h->changed_valid = false;
    h->ring_m = 0;
    return rebuild_view(h);  // loaded streams stay loaded (their indices are still valid); the per-round index is rebuilt
}

int rapid_view_is_safe_to_join(rapid_engine* h, int32_t node, int64_t id_hi, int64_t id_lo, int32_t* status_out) {
    int rc = check_node(h, node);
    if (rc) return rc;
    if (h->member[(size_t)node]) *status_out = RAPID_HOSTNAME_ALREADY_IN_RING;
    else {
        bool seen = false;
        if ((rc = use_device(h)) || (rc = ids_seen_any(h, {{id_hi, id_lo}}, &seen))) return rc;
        *status_out = seen ? RAPID_UUID_ALREADY_IN_RING : RAPID_SAFE_TO_JOIN;
    }
    return RAPID_OK;
}

int rapid_view_ring_add(rapid_engine* h, int32_t node, int64_t id_hi, int64_t id_lo) {
    int rc = check_node(h, node);
    if (rc) return rc;
    if ((rc = use_device(h))) return rc;
    const std::pair<int64_t, int64_t> id{id_hi, id_lo};
    bool seen = false;
    if ((rc = ids_seen_any(h, {id}, &seen))) return rc;
    if (seen) return fail(h, RAPID_EUUID_SEEN, "identifier of node %d already seen", node);  // :127-129
    if (h->member[(size_t)node]) return fail(h, RAPID_ENODE_EXISTS, "node %d already in ring", node);          // :133-135
    const int64_t old_hi = h->id_hi[(size_t)node], old_lo = h->id_lo[(size_t)node];
    const int n_ids_before = h->n_ids_dev;
    const size_t pending_before = h->ids_pending.size();
    h->member[(size_t)node] = 1;
    h->changed.push_back(node);
    h->id_hi[(size_t)node] = id_hi;
    h->id_lo[(size_t)node] = id_lo;
    h->ids_pending.push_back(id);
    if ((rc = rebuild_view(h))) {  // the node is not added: flags and identifier as before, the rings sorted afresh by the next change
        h->member[(size_t)node] = 0;
        h->id_hi[(size_t)node] = old_hi;
        h->id_lo[(size_t)node] = old_lo;
        if (h->ids_pending.size() > pending_before) h->ids_pending.resize(pending_before);
        view_change_failed(h, n_ids_before);
    }
    return rc;
}

int rapid_view_ring_delete(rapid_engine* h, int32_t node) {
    int rc = check_node(h, node);
    if (rc) return rc;
    if ((rc = use_device(h))) return rc;
    if (!h->member[(size_t)node]) return fail(h, RAPID_ENODE_MISSING, "node %d not in ring", node);  // :172-174
    h->member[(size_t)node] = 0;  // identifiersSeen is never pruned (:167-201)
    h->changed.push_back(node);
    if ((rc = rebuild_view(h))) {
        h->member[(size_t)node] = 1;
        view_change_failed(h, h->n_ids_dev);
    }
    return rc;
}

int rapid_view_observers(rapid_engine* h, int32_t node, int32_t* out, int32_t cap, int32_t* n_out) {
    int rc = check_node(h, node);
    if (rc) return rc;
    if (!h->member[(size_t)node]) return fail(h, RAPID_ENODE_MISSING, "node %d not in ring", node);
    if ((rc = use_device(h)) || (rc = ensure_host_tables(h))) return rc;
    const int K = h->cfg.K;
    return copy_list(h, h->h_obs.data() + (size_t)node * K, h->n_members <= 1 ? 0 : K, out, cap, n_out);
}

int rapid_view_subjects(rapid_engine* h, int32_t node, int32_t* out, int32_t cap, int32_t* n_out) {
    int rc = check_node(h, node);
    if (rc) return rc;
    if (!h->member[(size_t)node]) return fail(h, RAPID_ENODE_MISSING, "node %d not in ring", node);
    if ((rc = use_device(h)) || (rc = ensure_host_tables(h))) return rc;
    const int K = h->cfg.K;
    return copy_list(h, h->h_subj.data() + (size_t)node * K, h->n_members <= 1 ? 0 : K, out, cap, n_out);
}

int rapid_view_expected_observers(rapid_engine* h, int32_t node, int32_t* out, int32_t cap, int32_t* n_out) {
    int rc = check_node(h, node);
    if (rc) return rc;
    if ((rc = use_device(h)) || (rc = ensure_host_tables(h))) return rc;
    const int K = h->cfg.K;
    if (h->n_members == 0) return copy_list(h, nullptr, 0, out, cap, n_out);  // :296-298
    if (!h->member[(size_t)node]) return copy_list(h, h->h_obs.data() + (size_t)node * K, K, out, cap, n_out);
    // a member's "expected observers" are its predecessors (:299 -> :308-322); alone in the ring it precedes itself
    if (h->n_members == 1) {
        std::vector<int> self((size_t)K, node);
        return copy_list(h, self.data(), K, out, cap, n_out);
    }
    return copy_list(h, h->h_subj.data() + (size_t)node * K, K, out, cap, n_out);
}

int rapid_view_ring_numbers(rapid_engine* h, int32_t observer, int32_t subject, int32_t* out, int32_t cap,
                            int32_t* n_out) {
    int rc = check_node(h, observer);
    if (rc) return rc;
    if (!h->member[(size_t)observer]) return fail(h, RAPID_ENODE_MISSING, "node %d not in ring", observer);
    if ((rc = use_device(h)) || (rc = ensure_host_tables(h))) return rc;
    const int K = h->cfg.K;
    std::vector<int> idx;
    if (h->n_members > 1)
        for (int k = 0; k < K; ++k)
            if (h->h_subj[(size_t)observer * K + k] == subject) idx.push_back(k);
    return copy_list(h, idx.data(), (int)idx.size(), out, cap, n_out);
}

int rapid_view_ring(rapid_engine* h, int32_t k, int32_t* out, int32_t cap, int32_t* n_out) {
    if (!h || !h->view_built) return fail(h, RAPID_ESTATE, "view not built");
    if (k < 0 || k >= h->cfg.K) return fail(h, RAPID_EINVAL, "ring %d out of range", k);
    int rc;
    if ((rc = use_device(h)) || (rc = ensure_host_tables(h))) return rc;
    return copy_list(h, h->h_ring.data() + (size_t)k * h->n_members, h->n_members, out, cap, n_out);
}

int rapid_view_ring_key(rapid_engine* h, int32_t k, int32_t node, int64_t* key_out) {
    int rc = check_node(h, node);
    if (rc) return rc;
    if (k < 0 || k >= h->cfg.K) return fail(h, RAPID_EINVAL, "ring %d out of range", k);
    if ((rc = use_device(h)) || (rc = ensure_host_tables(h))) return rc;
    *key_out = h->h_keys[(size_t)k * h->n_nodes + node];
    return RAPID_OK;
}

int rapid_view_is_host_present(rapid_engine* h, int32_t node, int32_t* present_out) {
    int rc = check_node(h, node);
    if (rc) return rc;
    *present_out = h->member[(size_t)node] ? 1 : 0;
    return RAPID_OK;
}

int rapid_view_size(rapid_engine* h, int32_t* n_out) {
    if (!h || !h->view_built) return fail(h, RAPID_ESTATE, "view not built");
    *n_out = h->n_members;
    return RAPID_OK;
}

int rapid_view_config_id(rapid_engine* h, int64_t* id_out) {
    if (!h || !h->view_built) return fail(h, RAPID_ESTATE, "view not built");
    *id_out = h->config_id;
    return RAPID_OK;
}

int rapid_view_q4_at_risk(rapid_engine* h, const int32_t* hot, int32_t n, int32_t* out, int32_t cap, int32_t* n_out) {
    if (!h || n < 0 || (n > 0 && !hot) || cap < 0 || (cap > 0 && !out) || !n_out) return RAPID_EINVAL;
    if (!h->view_built) return fail(h, RAPID_ESTATE, "view not built");
    int rc = use_device(h);
    if (rc) return rc;
    std::vector<int> nodes;
    nodes.reserve((size_t)n);
    for (int i = 0; i < n; ++i) {
        if (hot[i] < 0 || hot[i] >= h->n_nodes) return fail(h, RAPID_EINVAL, "node index %d out of range", hot[i]);
        if (h->member[(size_t)hot[i]]) nodes.push_back(hot[i]);
    }
    std::sort(nodes.begin(), nodes.end());
    nodes.erase(std::unique(nodes.begin(), nodes.end()), nodes.end());
    *n_out = 0;
    // (a view of one member has no observers: computeObserversOf returns the empty list, :240-242 -- nothing to go stale)
    if (nodes.empty() || h->n_members <= 1) return RAPID_OK;
    const size_t m = nodes.size();
    std::vector<unsigned char> flag(m);
    HIPCHK(h, h->d_q4_nodes.ensure(m));
    HIPCHK(h, h->d_q4_flag.ensure(m));
    HIPCHK(h, hipMemcpyAsync(h->d_q4_nodes.p, nodes.data(), sizeof(int) * m, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(rapid::q4_check_kernel, dim3(grid_for((long long)m, 256)), dim3(256), 0, h->stream, h->d_q4_nodes.p, (int)m, h->d_member.p, h->d_obs.p,
                       h->cfg.K, h->d_q4_rows.p, h->d_q4_valid.p, h->d_q4_flag.p);
    HIPCHK(h, hipMemcpyAsync(flag.data(), h->d_q4_flag.p, m, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    int at_risk = 0;
    for (size_t i = 0; i < m; ++i)
        if (flag[i]) {
            if (at_risk < cap) out[at_risk] = nodes[i];
            ++at_risk;
        }
    *n_out = at_risk;
    if (at_risk > cap) return fail(h, RAPID_ECAPACITY, "%d subjects at risk, capacity %d", at_risk, cap);
    return RAPID_OK;
}

int rapid_view_q4_emulation(rapid_engine* h, int32_t on) {
    if (!h) return RAPID_EINVAL;
    if (h->q4_emulate != (on != 0)) h->index_valid = false;
    h->q4_emulate = on != 0;
    return RAPID_OK;
}

int rapid_view_tables(rapid_engine* h, int32_t* observers, int32_t* subjects, uint8_t* member, int32_t n_nodes) {
    if (!h || !h->view_built) return fail(h, RAPID_ESTATE, "view not built");
    if (n_nodes != h->n_nodes) return fail(h, RAPID_EINVAL, "n_nodes mismatch (%d vs %d)", n_nodes, h->n_nodes);
    int rc;
    if ((rc = use_device(h)) || (rc = ensure_host_tables(h))) return rc;
    const size_t n = (size_t)h->cfg.K * n_nodes;
    if (observers) std::memcpy(observers, h->h_obs.data(), n * sizeof(int));
    if (subjects) std::memcpy(subjects, h->h_subj.data(), n * sizeof(int));
    if (member) std::memcpy(member, h->member.data(), (size_t)n_nodes);
    return RAPID_OK;
}

// ------------------------------------------------------------------------------------ single detector
int rapid_cd_create(rapid_engine* h, int32_t K, int32_t H, int32_t L, rapid_cd** out) {
    if (!h || !out) return RAPID_EINVAL;
    *out = nullptr;
    if (!valid_khl(K, H, L) || K > RAPID_MAX_K) return fail(h, RAPID_EINVAL, "Arguments do not satisfy K > H >= L >= 0");
    int rc = use_device(h);
    if (rc) return rc;
    rapid_cd* cd = new rapid_cd();
    cd->eng = h;
    cd->K = K;
    cd->H = H;
    cd->L = L;
    cd->n_nodes = h->cfg.n_max;
    const size_t n_padded = (size_t)((cd->n_nodes + 7) / 8) * 8;
    hipError_t e = cd->d_state.ensure(n_padded);
    if (e == hipSuccess) e = cd->d_scal.ensure(4);
    if (e == hipSuccess) e = cd->d_out_n.ensure(1);
    if (e == hipSuccess) e = cd->d_out.ensure((size_t)cd->n_nodes);
    if (e == hipSuccess) e = hipMemsetAsync(cd->d_state.p, 0, n_padded * 2, h->stream);
    if (e == hipSuccess) e = hipMemsetAsync(cd->d_scal.p, 0, 16, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) {
        delete cd;
        return fail(h, RAPID_EDEVICE, "rapid_cd_create: %s", hipGetErrorString(e));
    }
    *out = cd;
    return RAPID_OK;
}

void rapid_cd_destroy(rapid_cd* cd) {
    if (!cd) return;
    (void)hipSetDevice(cd->eng->cfg.device_id);
    cd->d_state.release(); cd->d_scal.release(); cd->d_out.release(); cd->d_counts.release();
    cd->d_out_n.release(); cd->d_alerts.release();
    delete cd;
}

static int cd_run(rapid_cd* cd, const rapid_alert_record* alerts, int n, int mode, int32_t* out_idx, int32_t cap,
                  int32_t* out_counts, int32_t* n_out) {
    rapid_engine* h = cd->eng;
    int rc = use_device(h);
    if (rc) return rc;
    if (mode == 1) {
        if (!h->view_built) return fail(h, RAPID_ESTATE, "invalidateFailingEdges needs a view");
        if (cd->K != h->cfg.K) return fail(h, RAPID_EINVAL, "detector K=%d differs from the view's K=%d", cd->K, h->cfg.K);
    }
    HIPCHK(h, cd->d_counts.ensure((size_t)std::max(n, 1)));
    HIPCHK(h, cd->d_alerts.ensure((size_t)std::max(n, 1) * 20));
    if (n) HIPCHK(h, hipMemcpyAsync(cd->d_alerts.p, alerts, (size_t)n * 20, hipMemcpyHostToDevice, h->stream));
    rapid::CdParams p;
    p.state = cd->d_state.p;
    p.scal = cd->d_scal.p;
    p.alerts = cd->d_alerts.p;
    p.n_alerts = n;
    p.n_nodes = mode == 1 ? std::min(cd->n_nodes, h->n_nodes) : cd->n_nodes;
    p.K = cd->K;
    p.H = cd->H;
    p.L = cd->L;
    p.obs = h->view_built ? h->d_obs.p : nullptr;
    p.out_idx = cd->d_out.p;
    p.out_cap = cd->n_nodes;
    p.out_counts = cd->d_counts.p;
    p.out_n = cd->d_out_n.p;
    p.mode = mode;
    hipLaunchKernelGGL(rapid::cd_instance_kernel, dim3(1), dim3(64), 0, h->stream, p);
    int total = 0;
    HIPCHK(h, hipMemcpyAsync(&total, cd->d_out_n.p, 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipGetLastError());
    if (n_out) *n_out = total;
    if (out_counts && n) HIPCHK(h, hipMemcpy(out_counts, cd->d_counts.p, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost));
    if (total > cap) return fail(h, RAPID_ECAPACITY, "proposal of %d nodes exceeds capacity %d", total, cap);
    if (total) HIPCHK(h, hipMemcpy(out_idx, cd->d_out.p, sizeof(int) * (size_t)total, hipMemcpyDeviceToHost));
    return RAPID_OK;
}

int rapid_cd_aggregate(rapid_cd* cd, const rapid_alert_record* alerts, int32_t n, int32_t* out_idx, int32_t cap,
                       int32_t* out_counts, int32_t* n_out) {
    if (!cd || n < 0 || (n > 0 && !alerts)) return RAPID_EINVAL;
    return cd_run(cd, alerts, n, 0, out_idx, cap, out_counts, n_out);
}

int rapid_cd_invalidate(rapid_cd* cd, int32_t* out_idx, int32_t cap, int32_t* n_out) {
    if (!cd) return RAPID_EINVAL;
    return cd_run(cd, nullptr, 0, 1, out_idx, cap, nullptr, n_out);
}

int rapid_cd_num_proposals(rapid_cd* cd, int32_t* n_out) {
    if (!cd || !n_out) return RAPID_EINVAL;
    rapid_engine* h = cd->eng;
    int rc = use_device(h);
    if (rc) return rc;
    int scal[4];
    HIPCHK(h, hipMemcpyAsync(scal, cd->d_scal.p, 16, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    *n_out = scal[1];
    return RAPID_OK;
}

int rapid_cd_clear(rapid_cd* cd) {
    if (!cd) return RAPID_EINVAL;
    rapid_engine* h = cd->eng;
    int rc = use_device(h);
    if (rc) return rc;
    const size_t n_padded = (size_t)((cd->n_nodes + 7) / 8) * 8;
    HIPCHK(h, hipMemsetAsync(cd->d_state.p, 0, n_padded * 2, h->stream));
    HIPCHK(h, hipMemsetAsync(cd->d_scal.p, 0, 16, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return RAPID_OK;
}

// ------------------------------------------------------------------------------------------ population
// The delivered streams stay what they are when they cross the boundary: 20-byte records, read once per round by the tally
// kernel itself (tally_kernel.h: kFmtBoundary).  Loading = making them resident: a copy into the engine's own buffer (host
// records: the PCIe transfer; device records: a device-to-device copy, because the caller's buffer is only borrowed for the
// call), or no copy at all when the caller leaves them in place (rapid_sim_attach_streams_device).
static void streams_replaced(rapid_engine* h, int n_receivers, long long n_rec) {
    h->out_base = 0;
    h->tiled_total = 0;
    h->tiled_block = nullptr;
    h->streams_generated = false;
    h->n_receivers = n_receivers;
    h->n_records_total = n_rec;
    h->n_alert_set = -1;
    h->trust_copies = false;
    h->index_valid = false;
    h->streams_loaded = true;
    h->tallied = false;
    h->have_decision = false;
    h->tally_votes_valid = false;
    h->tally_settled_valid = false;
}

static int own_records(rapid_engine* h, const unsigned char* src, hipMemcpyKind kind, long long n_rec) {
    const size_t bytes = (size_t)n_rec * 20;
    HIPCHK(h, h->d_records_own.ensure(bytes + 64));
    if (bytes) HIPCHK(h, hipMemcpyAsync(h->d_records_own.p, src, bytes, kind, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));  // (borrowed for the call)
    h->d_records = h->d_records_own.p;
    h->records_bytes = bytes;
    h->rec_fmt = rapid::kFmtBoundary;
    h->offsets_on_device_only = false;
    return RAPID_OK;
}

int rapid_sim_load_streams(rapid_engine* h, const rapid_alert_record* records, const int64_t* rec_off,
                           int32_t n_receivers) {
    if (!h || !rec_off || n_receivers < 0) return RAPID_EINVAL;
    int rc = use_device(h);
    if (rc) return rc;
    const long long n_rec = rec_off[n_receivers];
    if (rec_off[0] != 0 || n_rec < 0 || (n_rec > 0 && !records)) return fail(h, RAPID_EINVAL, "bad record stream");
    for (int r = 0; r < n_receivers; ++r) {
        if (rec_off[r + 1] < rec_off[r]) return fail(h, RAPID_EINVAL, "rec_off not monotone at %d", r);
        if (rec_off[r + 1] - rec_off[r] > rapid::kMaxStreamRecords)
            return fail(h, RAPID_ECAPACITY, "receiver %d: %lld records; at most %lld per stream", r, (long long)(rec_off[r + 1] - rec_off[r]), (long long)rapid::kMaxStreamRecords);
    }
    HIPCHK(h, h->d_rec_off_own.ensure((size_t)n_receivers + 1));
    HIPCHK(h, hipMemcpyAsync(h->d_rec_off_own.p, rec_off, sizeof(long long) * ((size_t)n_receivers + 1), hipMemcpyHostToDevice,
                             h->stream));
    h->streams_loaded = false;  // (until the records are in place: a failing copy leaves no half-loaded streams behind)
    if ((rc = own_records(h, reinterpret_cast<const unsigned char*>(records), hipMemcpyHostToDevice, n_rec))) return rc;
    h->d_rec_off = h->d_rec_off_own.p;
    streams_replaced(h, n_receivers, n_rec);
    return RAPID_OK;
}

// d_rec_off as the caller's device array: its last entry and the per-stream bound are checked on a host copy
static int check_device_offsets(rapid_engine* h, const int64_t* d_rec_off, int32_t n_receivers, uint64_t records_bytes, long long* n_rec_out) {
    std::vector<long long> off((size_t)n_receivers + 1);
    HIPCHK(h, hipMemcpy(off.data(), d_rec_off, sizeof(long long) * off.size(), hipMemcpyDeviceToHost));
    const long long n_rec = off[(size_t)n_receivers];
    if (off[0] != 0 || n_rec < 0 || (unsigned long long)n_rec * 20ull > records_bytes)
        return fail(h, RAPID_EINVAL, "records_bytes=%llu does not cover %lld records", (unsigned long long)records_bytes, n_rec);
    for (int r = 0; r < n_receivers; ++r) {
        if (off[(size_t)r + 1] < off[(size_t)r]) return fail(h, RAPID_EINVAL, "rec_off not monotone at %d", r);
        if (off[(size_t)r + 1] - off[(size_t)r] > rapid::kMaxStreamRecords)
            return fail(h, RAPID_ECAPACITY, "receiver %d: %lld records; at most %lld per stream", r, off[(size_t)r + 1] - off[(size_t)r], (long long)rapid::kMaxStreamRecords);
    }
    *n_rec_out = n_rec;
    return RAPID_OK;
}

int rapid_sim_load_streams_device(rapid_engine* h, const void* d_records, uint64_t records_bytes,
                                  const int64_t* d_rec_off, int32_t n_receivers) {
    if (!h || !d_rec_off || n_receivers < 0 || (!d_records && records_bytes)) return RAPID_EINVAL;
    int rc = use_device(h);
    if (rc) return rc;
    if ((rc = check_borrowed(h, d_records, records_bytes, "rapid_sim_load_streams_device: records")) ||
        (rc = check_borrowed(h, d_rec_off, 8ull * ((unsigned long long)n_receivers + 1ull), "rapid_sim_load_streams_device: offsets")))
        return rc;
    long long n_rec = 0;
    if ((rc = check_device_offsets(h, d_rec_off, n_receivers, records_bytes, &n_rec))) return rc;
    h->streams_loaded = false;
    if ((rc = own_records(h, static_cast<const unsigned char*>(d_records), hipMemcpyDeviceToDevice, n_rec))) return rc;
    h->d_rec_off = reinterpret_cast<const long long*>(d_rec_off);  // the offsets stay where they are (borrowed until the next load)
    streams_replaced(h, n_receivers, n_rec);
    return RAPID_OK;
}

int rapid_sim_attach_streams_device(rapid_engine* h, const void* d_records, uint64_t records_bytes, const int64_t* d_rec_off,
                                    int32_t n_receivers) {
    if (!h || !d_rec_off || n_receivers < 0 || (!d_records && records_bytes)) return RAPID_EINVAL;
    if ((reinterpret_cast<uintptr_t>(d_records) & 3u) != 0u) return fail(h, RAPID_EINVAL, "records must be 4-byte aligned");
    int rc = use_device(h);
    if (rc) return rc;
    if ((rc = check_borrowed(h, d_records, records_bytes, "rapid_sim_attach_streams_device: records")) ||
        (rc = check_borrowed(h, d_rec_off, 8ull * ((unsigned long long)n_receivers + 1ull), "rapid_sim_attach_streams_device: offsets")))
        return rc;
    // On the round's path: nothing is copied, nothing is launched, nothing is waited for.  The offsets are checked where they are,
    // by the wave of the tally kernel that is about to follow them (tally_kernel.h: TallyParams::stream_bytes): offsets that do
    // not lie inside the records are not followed, and the round's results come back as RAPID_EINVAL (rapid_sim_results /
    // rapid_sim_count_votes / rapid_sim_round).
    h->d_records = static_cast<const unsigned char*>(d_records);  // the tally reads them in place
    h->records_bytes = (unsigned long long)records_bytes;
    h->rec_fmt = rapid::kFmtBoundary;
    h->d_rec_off = reinterpret_cast<const long long*>(d_rec_off);
    h->offsets_on_device_only = true;
    streams_replaced(h, n_receivers, (long long)(records_bytes / 20));  // (an upper bound until somebody needs the count: total_records())
    return RAPID_OK;
}

// The number of delivered records of the loaded streams: known on the host, except for an attached set whose offsets were
// never copied -- fetched (one synchronising 8-byte copy) by the few paths that need it: an undeclared alert set, the probes.
static int total_records(rapid_engine* h, long long* n);
namespace {
int total_records_fwd(rapid_engine* h, long long* n) { return total_records(h, n); }
}  // namespace
static int total_records(rapid_engine* h, long long* n) {
    if (h->offsets_on_device_only) {
        long long last = 0;
        HIPCHK(h, hipMemcpyAsync(&last, h->d_rec_off + h->n_receivers, 8, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (last < 0 || (unsigned long long)last * 20ull > h->records_bytes)
            return fail(h, RAPID_EINVAL, "the attached stream offsets run past records_bytes");
        h->n_records_total = last;
        h->offsets_on_device_only = false;
    }
    *n = h->n_records_total;
    return RAPID_OK;
}

int rapid_sim_generate(rapid_engine* h, const rapid_alert_record* alerts, const int64_t* batch_off, int32_t n_batches, const uint32_t* batch_keep,
                       const int32_t* receivers, int32_t n_receivers, uint64_t seed, int32_t format) {
    if (!h || !batch_off || n_batches < 0 || n_receivers < 0 || (n_receivers > 0 && !receivers)) return RAPID_EINVAL;
    if (format != RAPID_GEN_RESOLVED && format != RAPID_GEN_BOUNDARY) return fail(h, RAPID_EINVAL, "unknown record format %d", format);
    if (!h->view_built) return fail(h, RAPID_ESTATE, "view not built");
    int rc = use_device(h);
    if (rc) return rc;
    const long long A = batch_off[n_batches];
    if (batch_off[0] != 0 || A < 0 || (A > 0 && !alerts)) return fail(h, RAPID_EINVAL, "bad batch offsets");
    // (a BatchedAlertMessage is never empty: the reference's batcher only sends what it has queued, R/MembershipService.java:613-637;
    // an empty batch would still end -- invalidateFailingEdges runs once per message, :330 -- and has no record to say so)
    for (int b = 0; b < n_batches; ++b)
        if (batch_off[b + 1] <= batch_off[b]) return fail(h, RAPID_EINVAL, "batch %d is empty or the offsets decrease", b);
    if (A > rapid::kMaxStreamRecords) return fail(h, RAPID_ECAPACITY, "%lld alerts per receiver; at most %lld", A, (long long)rapid::kMaxStreamRecords);
    const long long total = (long long)n_receivers * A;
    const bool boundary = format == RAPID_GEN_BOUNDARY;
    hipStream_t st = h->stream;
    if (!h->ev0) {
        HIPCHK(h, hipEventCreate(&h->ev0));
        HIPCHK(h, hipEventCreate(&h->ev1));
    }
    // what the set itself says about the deliveries that will be copies of it (R/MembershipService.java:653-657)
    bool clean = true;
    for (long long i = 0; i < A; ++i) clean = clean && alerts[i].cfg_id == h->config_id && alerts[i].dst < (uint32_t)h->n_nodes;
    // Nothing of the engine's stream state is committed before the generation has succeeded: a failure leaves "no streams loaded".
    h->streams_loaded = false;
    h->index_valid = false;
    h->tallied = false;
    HIPCHK(h, h->d_alert_set.ensure((size_t)std::max<long long>(A, 1) * 20 + 16));
    HIPCHK(h, h->d_gen_boff.ensure((size_t)n_batches + 1));
    HIPCHK(h, h->d_gen_rx.ensure((size_t)std::max(n_receivers, 1)));
    if (batch_keep) HIPCHK(h, h->d_gen_keep.ensure((size_t)std::max(n_batches, 1)));
    if (A) HIPCHK(h, hipMemcpyAsync(h->d_alert_set.p, alerts, (size_t)A * 20, hipMemcpyHostToDevice, st));
    HIPCHK(h, hipMemcpyAsync(h->d_gen_boff.p, batch_off, sizeof(long long) * ((size_t)n_batches + 1), hipMemcpyHostToDevice, st));
    if (n_receivers) HIPCHK(h, hipMemcpyAsync(h->d_gen_rx.p, receivers, sizeof(int) * (size_t)n_receivers, hipMemcpyHostToDevice, st));
    if (batch_keep && n_batches) HIPCHK(h, hipMemcpyAsync(h->d_gen_keep.p, batch_keep, sizeof(unsigned int) * (size_t)n_batches, hipMemcpyHostToDevice, st));
    HIPCHK(h, hipStreamSynchronize(st));  // (borrowed inputs)
    const size_t stride = boundary ? 20 : 8;
    HIPCHK(h, h->d_records_own.ensure((size_t)total * stride + 64));
    HIPCHK(h, h->d_rec_off_own.ensure((size_t)n_receivers + 1));
    hipLaunchKernelGGL(rapid::gen_offsets_kernel, dim3(grid_for((long long)n_receivers + 1, 256)), dim3(256), 0, st, h->d_rec_off_own.p, n_receivers, A);
    h->n_receivers = n_receivers;
    h->n_records_total = total;
    h->n_alert_set = A;
    h->d_alerts = h->d_alert_set.p;
    h->rec_fmt = boundary ? rapid::kFmtBoundary : rapid::kFmtResident;
    HIPCHK(h, h->d_errflags.ensure(2));
    HIPCHK(h, h->d_stats.ensure(stats_words(h)));
    HIPCHK(h, h->d_voteback.ensure((10 * 8 + ((size_t)h->max_cut + 1) * sizeof(int) + 7) / 8));
    // resolved records carry entries of the round's index: built first, from the alert set
    if (!boundary && (rc = build_round_index(h))) return rc;
    HIPCHK(h, hipEventRecord(h->ev0, st));
    if (total > 0) {
        if (!boundary) {
            HIPCHK(h, h->d_gen_res.ensure((size_t)A));
            hipLaunchKernelGGL(rapid::gen_resolve_alerts_kernel, dim3(grid_for(A, 256)), dim3(256), 0, st, h->d_alert_set.p, A, (long long)h->config_id,
                               (unsigned int)h->n_nodes, h->d_entries.p, h->d_gen_res.p);
        }
        HIPCHK(h, h->d_gen_bat.ensure((size_t)std::max(n_batches, 1)));
        hipLaunchKernelGGL(rapid::gen_pack_batches_kernel, dim3(grid_for(n_batches, 256)), dim3(256), 0, st, h->d_gen_boff.p, n_batches,
                           boundary ? (const uint2*)nullptr : h->d_gen_res.p, batch_keep ? h->d_gen_keep.p : (const unsigned int*)nullptr, h->d_gen_bat.p);
        hipLaunchKernelGGL(rapid::gen_streams_kernel, dim3(grid_for(n_receivers, rapid::kGenWavesPerBlock)), dim3(rapid::kGenWavesPerBlock * 64), 0, st,
                           h->d_gen_res.p, h->d_alert_set.p, h->d_gen_bat.p, n_batches, batch_keep ? h->d_gen_keep.p : (const unsigned int*)nullptr,
                           h->d_gen_rx.p, n_receivers, A, (unsigned long long)seed, h->d_records_own.p, boundary ? 1 : 0);
    }
    HIPCHK(h, hipEventRecord(h->ev1, st));
    HIPCHK(h, hipStreamSynchronize(st));
    HIPCHK(h, hipGetLastError());
    (void)hipEventElapsedTime(&h->generate_ms, h->ev0, h->ev1);
    h->d_records = h->d_records_own.p;
    h->records_bytes = (unsigned long long)total * stride;
    h->d_rec_off = h->d_rec_off_own.p;
    h->offsets_on_device_only = false;
    h->gen_cfg_id = h->config_id;
    h->gen_clean = clean && batch_keep == nullptr;  // (an undelivered batch's places hold empty records: harmless, but not copies)
    const bool index_valid = h->index_valid;
    streams_replaced(h, n_receivers, total);
    h->n_alert_set = A;  // (streams_replaced forgets a declared set: this one is the streams' own)
    h->d_alerts = h->d_alert_set.p;
    h->streams_generated = true;
    h->index_valid = index_valid;  // (resolved: the index the entries come from; rapid_sim_new_round builds it again -- same numbering)
    return RAPID_OK;
}

#ifdef RAPID_TEST_BUILD
int rapid_debug_read_records(rapid_engine* h, int64_t first, int32_t n, uint32_t* subjects, uint32_t* core_words) {
    if (!h || first < 0 || n < 0 || !subjects || !core_words) return RAPID_EINVAL;
    int rc = use_device(h);
    if (rc) return rc;
    long long n_loaded = 0;  // (the delivered count, not the size of an attached buffer)
    if (h->streams_loaded && (rc = total_records(h, &n_loaded))) return rc;
    if (!h->streams_loaded || first + n > n_loaded) return fail(h, RAPID_EINVAL, "records [%lld, %lld) not loaded", (long long)first, (long long)first + n);
    const size_t stride = h->rec_fmt == rapid::kFmtBoundary ? 20 : 8;
    std::vector<unsigned char> raw((size_t)n * stride);
    if (n) HIPCHK(h, hipMemcpyAsync(raw.data(), h->d_records + (size_t)first * stride, raw.size(), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (int i = 0; i < n; ++i) {
        uint32_t w[5] = {0, 0, 0, 0, 0};
        std::memcpy(w, raw.data() + (size_t)i * stride, stride);
        if (h->rec_fmt == rapid::kFmtBoundary) {
            subjects[i] = w[3];
            core_words[i] = rapid::core_word(w[4]);
        } else {
            subjects[i] = w[0];  // the subject's dict_entry (tally_kernel.h)
            core_words[i] = w[1];
        }
    }
    return RAPID_OK;
}

#endif  // RAPID_TEST_BUILD

int rapid_sim_set_alert_set(rapid_engine* h, const rapid_alert_record* alerts, int64_t n_alerts) {
    if (!h || n_alerts < 0 || (n_alerts > 0 && !alerts)) return RAPID_EINVAL;
    if (!h->streams_loaded) return fail(h, RAPID_ESTATE, "load the streams first");
    if (h->rec_fmt == rapid::kFmtResident)
        return fail(h, RAPID_ESTATE, "generated deliveries are copies of the alert set they were generated from (rapid_sim_generate declares it)");
    int rc = use_device(h);
    if (rc) return rc;
    // A round's alert set is new every round, so this call is on the round's path: the borrowed records go through a pinned
    // staging buffer and an asynchronous copy on the engine's stream -- no stream synchronisation (the previous copy out of
    // the staging buffer, if it is still in flight, is waited for through its event).
    const size_t bytes = (size_t)n_alerts * 20;
    HIPCHK(h, h->d_alert_set.ensure(std::max<size_t>(bytes, 20) + 16));
    if (bytes > h->alert_stage_bytes) {
        if (h->h_alert_stage) {
            HIPCHK(h, hipStreamSynchronize(h->stream));
            (void)hipHostFree(h->h_alert_stage);
            h->h_alert_stage = nullptr;
            h->alert_stage_bytes = 0;
        }
        const size_t want = (std::max<size_t>(bytes + bytes / 4, 65536) + 65535) & ~(size_t)65535;  // (64 KiB granules, like the arena)
        HIPCHK(h, hipHostMalloc(reinterpret_cast<void**>(&h->h_alert_stage), want, hipHostMallocMapped | hipHostMallocCoherent));
        h->alert_stage_bytes = want;
    }
    if (!h->ev_alert) HIPCHK(h, hipEventCreateWithFlags(&h->ev_alert, hipEventDisableTiming));
    if (bytes) {
        if (h->alert_copy_pending) HIPCHK(h, hipEventSynchronize(h->ev_alert));
        std::memcpy(h->h_alert_stage, alerts, bytes);
        HIPCHK(h, hipMemcpyAsync(h->d_alert_set.p, h->h_alert_stage, bytes, hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipEventRecord(h->ev_alert, h->stream));
        h->alert_copy_pending = true;
    }
    h->n_alert_set = n_alerts;
    h->d_alerts = h->d_alert_set.p;
    h->index_valid = false;
    // the delivered records were generated from ANOTHER set: what the library could vouch for ("every record is a copy of a declared
    // alert of the current configuration") it can vouch for no longer -- records_known_current() must look at the records again
    h->streams_generated = false;
    return RAPID_OK;
}

int rapid_sim_set_alert_set_device(rapid_engine* h, const void* d_alerts, uint64_t alerts_bytes, int64_t n_alerts) {
    if (!h || n_alerts < 0 || (n_alerts > 0 && !d_alerts)) return RAPID_EINVAL;
    if ((reinterpret_cast<uintptr_t>(d_alerts) & 3u) != 0u) return fail(h, RAPID_EINVAL, "alerts must be 4-byte aligned");
    if ((unsigned long long)n_alerts > alerts_bytes / 20ull)  // (the index kernel would read past the caller's allocation)
        return fail(h, RAPID_EINVAL, "alerts_bytes=%llu does not cover %lld alerts", (unsigned long long)alerts_bytes, (long long)n_alerts);
    if (!h->streams_loaded) return fail(h, RAPID_ESTATE, "load the streams first");
    if (h->rec_fmt == rapid::kFmtResident)
        return fail(h, RAPID_ESTATE, "generated deliveries are copies of the alert set they were generated from (rapid_sim_generate declares it)");
    {
        int rc = use_device(h);
        if (rc) return rc;
        if ((rc = check_borrowed(h, d_alerts, (unsigned long long)n_alerts * 20ull, "rapid_sim_set_alert_set_device: alerts"))) return rc;
    }
    h->d_alerts = static_cast<const unsigned char*>(d_alerts);  // read in place by the round index; nothing is copied or waited for
    h->n_alert_set = n_alerts;
    h->index_valid = false;
    h->streams_generated = false;  // (as in rapid_sim_set_alert_set)
    return RAPID_OK;
}

int rapid_sim_trust_alert_copies(rapid_engine* h, int32_t on) {
    if (!h) return RAPID_EINVAL;
    if (on < 0 || on > 2) return RAPID_EINVAL;
    if (h->trust_copies != (on != 0)) h->index_valid = false;  // (another instantiation, maybe another launch geometry)
    h->trust_copies = on != 0;
    h->no_late_copies = on == 2;
    return RAPID_OK;
}

int rapid_sim_new_round(rapid_engine* h) {
    if (!h) return RAPID_EINVAL;
    if (!h->streams_loaded) return fail(h, RAPID_ESTATE, "no alert streams loaded");
    h->index_valid = false;  // rebuilt by the next tally, as after a load
    h->tallied = false;
    h->have_decision = false;
    h->tally_votes_valid = false;
    h->tally_settled_valid = false;
    return RAPID_OK;
}

int rapid_sim_tally(rapid_engine* h) {
    if (!h) return RAPID_EINVAL;
    int rc = use_device(h);
    if (rc) return rc;
    if (h->tiled_total) return fail(h, RAPID_ESTATE, "the last round was taken tile by tile: load, attach or generate streams first");
    h->tally_votes_valid = false;  // (set again by the launch below; a call that launches nothing must not leave an older launch's statistics valid)
    h->tally_settled_valid = false;
    if ((rc = prepare_tally(h))) return rc;
    if (!h->stats_fresh) HIPCHK(h, hipMemsetAsync(h->d_stats.p, 0, (size_t)64 * ((size_t)std::max(h->grid_blocks, 1) + 1), h->stream));
    if (h->n_receivers > 0) {
        if ((rc = launch_tally(h))) return rc;
        HIPCHK(h, hipGetLastError());
    }
    h->tallied = true;
    h->have_decision = false;
    return RAPID_OK;
}

// The tally kernel's sticky error word, read at the calls that synchronise with it anyway.
static int check_tally_errors(rapid_engine* h) {
    unsigned int flags[2] = {0u, 0u};
    HIPCHK(h, hipMemcpyAsync(flags, h->d_errflags.p, sizeof flags, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (flags[0] & 2u)
        return fail(h, RAPID_EINVAL, "a receiver's stream offsets do not lie inside the attached records (not ascending, longer than a stream's capacity, or past records_bytes)");
    if (flags[0] & 1u)
        return fail(h, RAPID_EINVAL, "a delivered alert names a subject / ring that the declared alert set does not contain "
                                    "(rapid_sim_set_alert_set must be given every distinct alert of the loaded streams)");
    return RAPID_OK;
}

int rapid_sim_results(rapid_engine* h, int32_t* emit_batch, int32_t* num_proposals, int32_t* prop_count,
                      uint64_t* fingerprint, int32_t n_receivers) {
    if (!h) return RAPID_EINVAL;
    if (!h->tallied) return fail(h, RAPID_ESTATE, "no tally has run");
    if (n_receivers != h->n_receivers) return fail(h, RAPID_EINVAL, "n_receivers mismatch");
    int rc = use_device(h);
    if (rc) return rc;
    const size_t R = (size_t)n_receivers;
    if (R) {
        if (emit_batch) HIPCHK(h, hipMemcpyAsync(emit_batch, h->d_emit.p, 4 * R, hipMemcpyDeviceToHost, h->stream));
        if (num_proposals) HIPCHK(h, hipMemcpyAsync(num_proposals, h->d_nprop.p, 4 * R, hipMemcpyDeviceToHost, h->stream));
        if (prop_count) HIPCHK(h, hipMemcpyAsync(prop_count, h->d_pcount.p, 4 * R, hipMemcpyDeviceToHost, h->stream));
        if (fingerprint) HIPCHK(h, hipMemcpyAsync(fingerprint, h->d_fp.p, 8 * R, hipMemcpyDeviceToHost, h->stream));
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipGetLastError());
    return check_tally_errors(h);
}

static void sort_ring0(rapid_engine* h, std::vector<int>& v) {
    // R/MembershipService.java:346-348: sorted(membershipView.getRingZeroComparator()) -- signed key compare
    // (keys gathered first: the comparisons then run on a few KB instead of chasing an 80 KB table per compare; equal keys
    // keep the order they came in, ascending node index, like a stable sort of the list)
    const long long* k0 = h->h_keys0.data();  // (ensure_host_keys0: every caller has called it)
    std::vector<std::pair<long long, int>> kv(v.size());
    for (size_t i = 0; i < v.size(); ++i) kv[i] = {k0[v[i]], (int)i};
    std::sort(kv.begin(), kv.end());
    std::vector<int> out(v.size());
    for (size_t i = 0; i < v.size(); ++i) out[i] = v[(size_t)kv[i].second];
    v.swap(out);
}

int rapid_sim_proposal(rapid_engine* h, int32_t receiver, int32_t* out, int32_t cap, int32_t* n_out) {
    if (!h) return RAPID_EINVAL;
    if (!h->tallied) return fail(h, RAPID_ESTATE, "no tally has run");
    if (receiver < 0 || receiver >= h->n_receivers) return fail(h, RAPID_EINVAL, "receiver out of range");
    int rc;
    if ((rc = use_device(h)) || (rc = ensure_host_keys0(h))) return rc;
    // (after a tiled round the node lists of the LAST tile are what is still resident; every receiver's counts and fingerprint are)
    int local = receiver;
    if (h->tiled_total) {
        if (receiver < h->tiled_last_base || receiver >= h->tiled_last_base + h->tiled_last_n)
            return fail(h, RAPID_ESTATE, "receiver %d's proposal went with its tile; receivers [%d, %d) of the last tile are resident", receiver,
                        h->tiled_last_base, h->tiled_last_base + h->tiled_last_n);
        local = receiver - h->tiled_last_base;
    }
    int cnt = 0;
    HIPCHK(h, hipMemcpyAsync(&cnt, h->d_pcount.p + receiver, 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (cnt < 0) return fail(h, RAPID_ECAPACITY, "receiver %d's proposal exceeds max_cut=%d", receiver, h->max_cut);
    std::vector<int> v((size_t)cnt);
    if (cnt)
        HIPCHK(h, hipMemcpy(v.data(), h->d_props.p + (size_t)local * h->max_cut, sizeof(int) * (size_t)cnt,
                            hipMemcpyDeviceToHost));
    sort_ring0(h, v);
    return copy_list(h, v.data(), cnt, out, cap, n_out);
}


// The host's reading of one answer block (res[10] + ref[]): fills *out and the engine's decision; kVoteNextSalt = two
// proposals share the winning bucket, count again with the next salt.
constexpr int kVoteNextSalt = 1;
static int decode_vote_answer(rapid_engine* h, const unsigned long long* hres, const int* href, rapid_round_result* out) {
    if ((unsigned int)hres[8] & 2u)
        return fail(h, RAPID_EINVAL, "a receiver's stream offsets do not lie inside the attached records (not ascending, longer than a stream's capacity, or past records_bytes)");
    if ((unsigned int)hres[8] & 1u)
        return fail(h, RAPID_EINVAL, "a delivered alert names a subject / ring that the declared alert set does not contain "
                                    "(rapid_sim_set_alert_set must be given every distinct alert of the loaded streams)");
    const unsigned long long* hw = hres;
    const unsigned long long* hmm = hres + 4;
    const unsigned long long* hmis = hres + 6;
    out->votes_total = (int64_t)hw[2];
    out->votes_winner = (int64_t)hw[1];
    out->distinct_local = (int32_t)hw[3];
    if (hw[1] == 0) return RAPID_OK;  // nobody proposed
    const unsigned long long fmax = hmm[0], fmin = ~hmm[1];
    if (fmax != fmin) {
        if ((long long)hw[1] < out->quorum) return RAPID_OK;  // no proposal can have a quorum
        return kVoteNextSalt;
    }
    const int ref_n = href[0];
    if (ref_n < 0 || ref_n == 0x7FFFFFFF) return fail(h, RAPID_ECAPACITY, "winning proposal exceeds max_cut=%d", h->max_cut);
    if (hmis[0] != 0 || hmis[1] != hw[1])
        return fail(h, RAPID_ECOLLISION, "fingerprint collision: %llu of %llu voters differ from the representative", hmis[0], hw[1]);
    out->cut_size = ref_n;
    // R/FastPaxos.java:146-150: |votesReceived| >= N - F and votes[proposal] >= N - F
    if ((long long)hw[1] >= out->quorum) {
        out->decided = 1;
        std::vector<int> ref(href + 1, href + 1 + ref_n);
        sort_ring0(h, ref);
        h->decided_cut = ref;
        h->have_decision = true;
    }
    return RAPID_OK;
}

static void begin_round_result(rapid_engine* h, rapid_round_result* out) {
    std::memset(out, 0, sizeof *out);
    const int N = h->n_members;
    const int F = (int)std::floor((double)(N - 1) / 4.0);  // R/FastPaxos.java:145
    out->membership_size = N;
    out->quorum = N - F;
    out->config_id = h->config_id;
    h->have_decision = false;
    h->decided_cut.clear();
}

int rapid_sim_count_votes(rapid_engine* h, rapid_round_result* out) {
    if (!h || !out) return RAPID_EINVAL;
    if (!h->tallied) return fail(h, RAPID_ESTATE, "no tally has run");
    if (h->tiled_total) return fail(h, RAPID_ESTATE, "the votes of a tiled round are counted, tile by tile, by rapid_sim_round_tiled itself");
    int rc;
    // (the ring-0 keys for the order of a decided cut: a host copy that outlives view changes -- mirroring every table of the
    // view here cost the first round after each view change 0.12 ms at 10^4 nodes, and 200 MB of copies at 10^6)
    if ((rc = use_device(h)) || (rc = ensure_host_keys0(h))) return rc;
    begin_round_result(h, out);
    hipStream_t st = h->stream;
    const int R = h->n_receivers;
    const size_t HB = (size_t)rapid::kVoteBuckets + 2;
    const size_t ref_len = (size_t)h->max_cut + 1;
    // One device buffer for everything the host reads back: res[8] = {winning bucket, its votes, voters, non-empty buckets,
    // max fingerprint, max ~fingerprint, mismatching voters, verified voters}, then ref[1 + max_cut]; one pinned copy of it.
    const size_t res_words = 10;  // ... + the tally kernel's sticky error word (res[8])
    const size_t back_bytes = res_words * 8 + ref_len * sizeof(int);
    HIPCHK(h, h->d_hist.ensure(HB));
    HIPCHK(h, h->d_winner.ensure(4));
    HIPCHK(h, h->d_mm.ensure(8));
    HIPCHK(h, h->d_voteback.ensure((back_bytes + 7) / 8));
    if ((rc = ensure_mailbox(h))) return rc;
    unsigned long long* const d_res = h->d_voteback.p;
    unsigned long long* const d_mismatch = d_res + 6;
    int* const d_ref = reinterpret_cast<int*>(d_res + res_words);
    unsigned long long* hres = h->h_pinned;
    const unsigned long long my_tag = ~(unsigned long long)h->rank;
    // a population held by one rank is counted by ONE workgroup (histogram in LDS) + the element-wise verification:
    // two launches, one copy, one synchronisation; larger ones and sharded ones go through the histogram in memory
    const bool local = !h->comm && R <= 262144;
    // a sharded population: every rank counts locally the same way, ONE all-gather exchanges the ranks' answers and every
    // rank merges them (vote_merge_kernel); only a round whose voters disagree goes through the histogram all-reduces
    bool merged = h->comm && R <= 262144 && (h->force_exact & 512) == 0;
    const size_t seg_words = (back_bytes + 7) / 8;
    if (merged) HIPCHK(h, h->d_gather.ensure(seg_words * (size_t)h->n_ranks));
    if ((local || merged) && !h->vote_lds_attr_set) {  // once per engine
        HIPCHK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(rapid::vote_count_local_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, rapid::kVoteBuckets * 4 + 1024));
        h->vote_lds_attr_set = true;
    }

    // Everything below is enqueued on the engine stream and read back with ONE synchronisation per salt:
    // histogram -> (all-reduce) -> winner -> min/max of the winning bucket -> (all-reduce) -> representative list
    // -> (all-reduce) -> element-wise verification -> (all-reduce).
    bool from_tally_used = false;
    unsigned int settled_seq = 0u;  // the sequence number the single-rank answer arrives under
    for (unsigned long long salt = 0; salt < 4; ++salt) {
        from_tally_used = false;
        if (merged) {
            // this rank's answer block: candidate, its verified votes, the rank's voters, the candidate's list -- from the
            // statistics the tally kernel gathered when they are fresh (no counting pass), else from the counting kernel
            const bool from_tally = h->tally_votes_valid && salt == 0 && (h->force_exact & 2048) == 0;
            const bool settled_by_tally = from_tally && h->tally_settled_valid;  // this rank's answer block is complete already (TallyParams::vote_cand)
            h->tally_votes_valid = false;
            h->tally_settled_valid = false;
            if (!from_tally)
                hipLaunchKernelGGL(rapid::vote_count_local_kernel, dim3(1), dim3(1024), (size_t)rapid::kVoteBuckets * 4 + 1024, st,
                                   h->d_fp.p, h->d_pcount.p, h->d_props.p, h->max_cut, R, salt, h->d_errflags.p, d_res, d_ref);
            // (published "to" the block itself: the last workgroup completes res[] and, from_tally, copies the list into ref[])
            const bool by_bits = from_tally && h->tally_bitmaps_valid && (h->force_exact & 262144) == 0;  // (the voters' bitmaps of the same launch: 64 bytes per receiver instead of a list)
            const int bits_wave = by_bits && h->bitmap_words > 16 ? 1 : 0;  // (bitmaps of more than 128 bytes: a wave per receiver)
            if (!settled_by_tally)
                hipLaunchKernelGGL(rapid::vote_verify_kernel, dim3(std::max(1u, grid_for((long long)R * (by_bits && !bits_wave ? 1 : 64), 1024))), dim3(1024), 0, st,
                                   h->d_fp.p, h->d_pcount.p, h->d_props.p, h->max_cut, R, d_res + 4, d_ref, d_mismatch, d_res,
                                   (int)res_words, reinterpret_cast<unsigned int*>(d_res + 9), reinterpret_cast<volatile unsigned long long*>(d_res),
                                   nullptr, 0u, from_tally ? 1 : 0, h->d_errflags.p, by_bits ? h->d_bitmaps.p : (const unsigned long long*)nullptr,
                                   h->bitmap_words, bits_wave);
            NCCLCHK(h, ncclAllGather(d_res, h->d_gather.p, seg_words, ncclUint64, h->comm, st));  // the round's one collective
            hipLaunchKernelGGL(rapid::vote_merge_kernel, dim3(1), dim3(256), 0, st, h->d_gather.p, h->n_ranks, (int)seg_words,
                               (int)res_words, h->max_cut, (long long)out->quorum, reinterpret_cast<volatile unsigned long long*>(h->d_mail + 64),
                               reinterpret_cast<volatile unsigned int*>(h->d_mail) + 14, ++h->mail_seq);
            HIPCHK(h, hipGetLastError());
            if ((rc = await_mail(h, 14, h->mail_seq))) return rc;
            if (reinterpret_cast<unsigned long long*>(h->h_mail + 64)[9] != 1ull) {  // the voters disagree somewhere: count the general way
                merged = false;
                --salt;
                continue;
            }
        } else if (local) {
            // the tally kernel gathered the voters' statistics itself (tally_kernel.h: vote_acc): if they are unanimous -- the
            // common round -- no counting pass runs at all; the verification below reads the representative's list in place
            const bool from_tally = h->tally_votes_valid && salt == 0 && (h->force_exact & 2048) == 0;
            // ... or the launch settled the round by itself (TallyParams::vote_cand): its last workgroup wrote the answer into the
            // host-mapped page under tally_settled_seq -- NOTHING is launched here, the host only waits for that word
            const bool settled_by_tally = from_tally && h->tally_settled_valid;
            h->tally_votes_valid = false;  // consumed: the verification below adds its counters to them
            h->tally_settled_valid = false;
            if (!from_tally)
                hipLaunchKernelGGL(rapid::vote_count_local_kernel, dim3(1), dim3(1024), (size_t)rapid::kVoteBuckets * 4 + 1024, st,
                                   h->d_fp.p, h->d_pcount.p, h->d_props.p, h->max_cut, R, salt, h->d_errflags.p, d_res, d_ref);
            from_tally_used = from_tally;
            const bool by_bits = from_tally && h->tally_bitmaps_valid && (h->force_exact & 262144) == 0;  // (the voters' bitmaps of the same launch: 64 bytes per receiver instead of a list)
            const int bits_wave = by_bits && h->bitmap_words > 16 ? 1 : 0;  // (bitmaps of more than 128 bytes: a wave per receiver)
            if (settled_by_tally)
                settled_seq = h->tally_settled_seq;
            else
                hipLaunchKernelGGL(rapid::vote_verify_kernel, dim3(std::max(1u, grid_for((long long)R * (by_bits && !bits_wave ? 1 : 64), 1024))), dim3(1024), 0, st,
                                   h->d_fp.p, h->d_pcount.p, h->d_props.p, h->max_cut, R, d_res + 4, d_ref, d_mismatch, d_res,
                                   (int)res_words, reinterpret_cast<unsigned int*>(d_res + 9),
                                   reinterpret_cast<volatile unsigned long long*>(h->d_mail + 64),
                                   reinterpret_cast<volatile unsigned int*>(h->d_mail) + 14, (settled_seq = ++h->mail_seq), from_tally ? 1 : 0,
                                   h->d_errflags.p, by_bits ? h->d_bitmaps.p : (const unsigned long long*)nullptr, h->bitmap_words, bits_wave);
        } else {
            HIPCHK(h, hipMemsetAsync(h->d_hist.p, 0, HB * 8, st));
            HIPCHK(h, hipMemsetAsync(h->d_mm.p, 0, 64, st));
            HIPCHK(h, hipMemsetAsync(d_res, 0, res_words * 8, st));
            if (R)
                hipLaunchKernelGGL(rapid::vote_histogram_kernel, dim3(grid_for(R, 256)), dim3(256), 0, st, h->d_fp.p, h->d_pcount.p,
                                   R, salt, h->d_hist.p);
            if (h->comm)  // the per-round all-reduce of the vote histogram over xGMI
                NCCLCHK(h, ncclAllReduce(h->d_hist.p, h->d_hist.p, HB, ncclUint64, ncclSum, h->comm, st));
            hipLaunchKernelGGL(rapid::vote_winner_kernel, dim3(1), dim3(1024), 0, st, h->d_hist.p, d_res);
            if (R)
                hipLaunchKernelGGL(rapid::vote_bucket_minmax_kernel, dim3(grid_for(R, 256)), dim3(256), 0, st, h->d_fp.p,
                                   h->d_pcount.p, R, salt, d_res, h->d_mm.p, my_tag);
            if (h->comm)  // mm[0..3] = {max fp, max ~fp, -, max ~rank of a rank holding a representative}
                NCCLCHK(h, ncclAllReduce(h->d_mm.p, h->d_mm.p, 4, ncclUint64, ncclMax, h->comm, st));
            // (a representative list too large for max_cut is reported as size INT_MAX, which survives the max-reduce
            // among the other ranks' zeros)
            hipLaunchKernelGGL(rapid::vote_prepare_ref_kernel, dim3(1), dim3(256), 0, st, h->d_mm.p, my_tag, h->comm ? 0 : 1,
                               h->d_pcount.p, h->d_props.p, h->max_cut, d_ref);
            if (h->comm) NCCLCHK(h, ncclAllReduce(d_ref, d_ref, ref_len, ncclInt32, ncclMax, h->comm, st));
            if (R)
                hipLaunchKernelGGL(rapid::vote_verify_kernel, dim3(grid_for((long long)R * 64, 1024)), dim3(1024), 0, st, h->d_fp.p,
                                   h->d_pcount.p, h->d_props.p, h->max_cut, R, h->d_mm.p, d_ref, d_mismatch, nullptr, 0, nullptr, nullptr, nullptr, 0u, 0, nullptr);
            if (h->comm) NCCLCHK(h, ncclAllReduce(d_mismatch, d_mismatch, 2, ncclUint64, ncclSum, h->comm, st));
            HIPCHK(h, hipMemcpyAsync(d_res + 4, h->d_mm.p, 16, hipMemcpyDeviceToDevice, st));
            HIPCHK(h, hipMemcpyAsync(d_res + 8, h->d_errflags.p, 8, hipMemcpyDeviceToDevice, st));
        }
        if (!local && !merged) HIPCHK(h, hipMemcpyAsync(hres, d_res, back_bytes, hipMemcpyDeviceToHost, st));
        HIPCHK(h, hipGetLastError());
        if (local) {
            if ((rc = await_mail(h, 14, settled_seq))) return rc;
            if (from_tally_used) {
                // settled iff the candidate (the lowest voter's proposal) has a quorum, or every voter holds it, or nobody
                // voted; otherwise the exact plurality count is owed: histogram pass, same salt again
                const unsigned long long* q = reinterpret_cast<unsigned long long*>(h->h_mail + 64);
                const unsigned long long votes = q[1], voters = q[2];
                if (!(voters == 0ull || votes == voters || (long long)votes >= (long long)out->quorum) && q[6] == 0ull) {
                    --salt;
                    continue;
                }
            }
        } else if (!merged) {
            HIPCHK(h, hipStreamSynchronize(st));
        }
        if (local || merged) {  // the last workgroup of the verification wrote the answer into host-mapped memory
            hres = reinterpret_cast<unsigned long long*>(h->h_mail + 64);
        }
        const int verdict = decode_vote_answer(h, hres, reinterpret_cast<const int*>(hres + res_words), out);
        if (verdict != kVoteNextSalt) return verdict;
    }
    return fail(h, RAPID_ECOLLISION, "winning vote bucket stayed impure under 4 salts");
}

int rapid_sim_decided_cut(rapid_engine* h, int32_t* out, int32_t cap, int32_t* n_out) {
    if (!h) return RAPID_EINVAL;
    if (!h->have_decision) return fail(h, RAPID_ESTATE, "no decision available");
    return copy_list(h, h->decided_cut.data(), (int)h->decided_cut.size(), out, cap, n_out);
}

#ifdef RAPID_TEST_BUILD
// Testing aids for the sharded count: what this engine's voters contribute to the all-gather, and the merge of n such
// contributions exactly as rapid_sim_count_votes runs it after its all-gather (one GPU stands in for n ranks).
int rapid_debug_vote_segment(rapid_engine* h, void* out, int64_t cap_bytes, int64_t* seg_bytes) {
    if (!h || !seg_bytes) return RAPID_EINVAL;
    if (!h->tallied) return fail(h, RAPID_ESTATE, "no tally has run");
    int rc;
    if ((rc = use_device(h))) return rc;
    const size_t res_words = 10, ref_len = (size_t)h->max_cut + 1;
    const size_t seg_words = (res_words * 8 + ref_len * sizeof(int) + 7) / 8;
    *seg_bytes = (int64_t)(seg_words * 8);
    if (!out) return RAPID_OK;
    if (cap_bytes < *seg_bytes) return fail(h, RAPID_ECAPACITY, "segment needs %lld bytes", (long long)*seg_bytes);
    if (h->tiled_total > 0 || h->tiled_block != nullptr) {  // a tiled round: the block its last pass left for the all-gather
        if (h->tiled_block == nullptr) return fail(h, RAPID_ESTATE, "the tiled round left no answer block");
        HIPCHK(h, hipMemcpyAsync(out, h->tiled_block, seg_words * 8, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        return RAPID_OK;
    }
    HIPCHK(h, h->d_voteback.ensure(seg_words));
    h->tally_votes_valid = false;
    h->tally_settled_valid = false;
    unsigned long long* const d_res = h->d_voteback.p;
    int* const d_ref = reinterpret_cast<int*>(d_res + res_words);
    const int R = h->n_receivers;
    hipStream_t st = h->stream;
    HIPCHK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(rapid::vote_count_local_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, rapid::kVoteBuckets * 4 + 1024));
    hipLaunchKernelGGL(rapid::vote_count_local_kernel, dim3(1), dim3(1024), (size_t)rapid::kVoteBuckets * 4 + 1024, st, h->d_fp.p,
                       h->d_pcount.p, h->d_props.p, h->max_cut, R, 0ull, h->d_errflags.p, d_res, d_ref);
    hipLaunchKernelGGL(rapid::vote_verify_kernel, dim3(std::max(1u, grid_for((long long)R * 64, 1024))), dim3(1024), 0, st, h->d_fp.p,
                       h->d_pcount.p, h->d_props.p, h->max_cut, R, d_res + 4, d_ref, d_res + 6, nullptr, 0, nullptr, nullptr, nullptr, 0u, 0, nullptr);
    HIPCHK(h, hipMemcpyAsync(out, d_res, seg_words * 8, hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipStreamSynchronize(st));
    return RAPID_OK;
}

int rapid_debug_vote_merge(rapid_engine* h, const void* segments, int32_t n_ranks, rapid_round_result* out, int32_t* status) {
    if (!h || !segments || n_ranks <= 0 || !out || !status) return RAPID_EINVAL;
    int rc;
    if ((rc = use_device(h)) || (rc = ensure_host_keys0(h)) || (rc = ensure_mailbox(h))) return rc;
    begin_round_result(h, out);
    const size_t res_words = 10, ref_len = (size_t)h->max_cut + 1;
    const size_t seg_words = (res_words * 8 + ref_len * sizeof(int) + 7) / 8;
    h->tiled_block = nullptr;  // (it lives in d_gather, which is about to be overwritten -- and maybe moved)
    HIPCHK(h, h->d_gather.ensure(seg_words * (size_t)n_ranks));
    hipStream_t st = h->stream;
    HIPCHK(h, hipMemcpyAsync(h->d_gather.p, segments, seg_words * 8 * (size_t)n_ranks, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(rapid::vote_merge_kernel, dim3(1), dim3(256), 0, st, h->d_gather.p, n_ranks, (int)seg_words, (int)res_words,
                       h->max_cut, (long long)out->quorum, reinterpret_cast<volatile unsigned long long*>(h->d_mail + 64), nullptr, 0u);
    HIPCHK(h, hipStreamSynchronize(st));
    HIPCHK(h, hipGetLastError());
    const unsigned long long* hres = reinterpret_cast<const unsigned long long*>(h->h_mail + 64);
    *status = (int32_t)hres[9];
    if (hres[9] != 1ull) return RAPID_OK;  // the voters disagree: rapid_sim_count_votes would count the general way
    const int verdict = decode_vote_answer(h, hres, reinterpret_cast<const int*>(hres + res_words), out);
    return verdict == kVoteNextSalt ? fail(h, RAPID_ESTATE, "merged answer is impure") : verdict;
}

#endif  // RAPID_TEST_BUILD

// ---- a round over a population that does not fit one launch ------------------------------------------------------------------
// (inside rapid_sim_round_tiled's tile loop: a failing call does not leave the generator running into buffers the caller may free)
#define HIPCHK_SYNC_GEN(expr)                                                                   \
    do {                                                                                        \
        const hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess) {                                                                 \
            if (h->stream_gen) (void)hipStreamSynchronize(h->stream_gen);                       \
            HIPCHK(h, e_);                                                                      \
        }                                                                                       \
    } while (0)

static int round_tiled_impl(rapid_engine* h, const rapid_alert_record* alerts, const int64_t* batch_off, int32_t n_batches,
                            const uint32_t* batch_keep, const int32_t* receivers, int32_t n_receivers, int32_t tile_receivers,
                            uint64_t seed, int32_t format, rapid_round_result* out);

// A tiled round that fails on the way -- a device error, a collective that returns an error, an impure vote bucket -- leaves NOTHING of
// its tile state behind: both streams are drained, the per-receiver results are invalid, no streams are loaded, and the caller's
// trust setting is what it was.  (Sharded: a rank that fails before the all-gather leaves its peers waiting in it -- the caller
// aborts the communicator, as after any rank failure)
int rapid_sim_round_tiled(rapid_engine* h, const rapid_alert_record* alerts, const int64_t* batch_off, int32_t n_batches,
                          const uint32_t* batch_keep, const int32_t* receivers, int32_t n_receivers, int32_t tile_receivers,
                          uint64_t seed, int32_t format, rapid_round_result* out) {
    const bool trust_before = h ? h->trust_copies : false, late_before = h ? h->no_late_copies : false;
    if (h) h->settle_in_tally = false;  // (the tiles' votes are accumulated across launches: vote_acc_pick / _count / _finish)
    const int rc = round_tiled_impl(h, alerts, batch_off, n_batches, batch_keep, receivers, n_receivers, tile_receivers, seed, format, out);
    if (h) {
        h->settle_in_tally = true;
        h->trust_copies = trust_before;  // (the round vouches for its own deliveries while it runs: see launch_tally)
        h->no_late_copies = late_before;
        h->out_base = 0;
        if (rc != RAPID_OK && rc != RAPID_EINVAL && rc != RAPID_ESTATE) {
            if (h->stream_gen) (void)hipStreamSynchronize(h->stream_gen);
            if (h->stream) (void)hipStreamSynchronize(h->stream);
            (void)hipGetLastError();
            h->n_receivers = 0;
            h->tiled_total = 0;
            h->tiled_block = nullptr;
            h->tiled_last_base = h->tiled_last_n = 0;
            h->tallied = false;
            h->have_decision = false;
            h->streams_loaded = false;
            h->streams_generated = false;
            h->index_valid = false;
        }
    }
    return rc;
}

static int round_tiled_impl(rapid_engine* h, const rapid_alert_record* alerts, const int64_t* batch_off, int32_t n_batches,
                            const uint32_t* batch_keep, const int32_t* receivers, int32_t n_receivers, int32_t tile_receivers,
                            uint64_t seed, int32_t format, rapid_round_result* out) {
    if (!h || !out || !batch_off || n_batches < 0 || n_receivers < 0 || (n_receivers > 0 && !receivers) || tile_receivers < 0) return RAPID_EINVAL;
    if (format != RAPID_GEN_RESOLVED && format != RAPID_GEN_BOUNDARY) return fail(h, RAPID_EINVAL, "unknown record format %d", format);
    if (!h->view_built) return fail(h, RAPID_ESTATE, "view not built");
    int rc = use_device(h);
    if (rc) return rc;
    const long long A = batch_off[n_batches];
    if (batch_off[0] != 0 || A < 0 || (A > 0 && !alerts)) return fail(h, RAPID_EINVAL, "bad batch offsets");
    for (int b = 0; b < n_batches; ++b)
        if (batch_off[b + 1] <= batch_off[b]) return fail(h, RAPID_EINVAL, "batch %d is empty or the offsets decrease", b);
    if (A > rapid::kMaxStreamRecords) return fail(h, RAPID_ECAPACITY, "%lld alerts per receiver; at most %lld", A, (long long)rapid::kMaxStreamRecords);
    if ((rc = ensure_host_keys0(h)) || (rc = ensure_mailbox(h))) return rc;
    const auto t_begin = std::chrono::steady_clock::now();
    const bool boundary = format == RAPID_GEN_BOUNDARY;
    const size_t stride = boundary ? 20 : 8;
    hipStream_t st = h->stream;
    const int R = n_receivers;
    // a tile = the receivers of one launch: sixteen waves' worth per CU unless the caller says otherwise, bounded by what 2^31 bytes
    // of boundary records hold (the tile's stream buffer is the only place a delivered record of this round ever exists)
    long long T = tile_receivers > 0 ? tile_receivers : (long long)16 * h->num_cus;
    T = std::max<long long>(1, std::min<long long>(T, std::max(R, 1)));
    bool clean = true;
    for (long long i = 0; i < A; ++i) clean = clean && alerts[i].cfg_id == h->config_id && alerts[i].dst < (uint32_t)h->n_nodes;
    // nothing of the engine's stream state survives into the round, and a failure on the way leaves "no streams loaded"
    h->streams_loaded = false;
    h->index_valid = false;
    h->tallied = false;
    h->have_decision = false;
    h->tiled_total = 0;
    h->tiled_block = nullptr;
    h->out_base = 0;
    HIPCHK(h, h->d_alert_set.ensure((size_t)std::max<long long>(A, 1) * 20 + 16));
    HIPCHK(h, h->d_gen_boff.ensure((size_t)n_batches + 1));
    HIPCHK(h, h->d_gen_rx.ensure((size_t)std::max(R, 1)));
    if (batch_keep) HIPCHK(h, h->d_gen_keep.ensure((size_t)std::max(n_batches, 1)));
    if (A) HIPCHK(h, hipMemcpyAsync(h->d_alert_set.p, alerts, (size_t)A * 20, hipMemcpyHostToDevice, st));
    HIPCHK(h, hipMemcpyAsync(h->d_gen_boff.p, batch_off, sizeof(long long) * ((size_t)n_batches + 1), hipMemcpyHostToDevice, st));
    if (R) HIPCHK(h, hipMemcpyAsync(h->d_gen_rx.p, receivers, sizeof(int) * (size_t)R, hipMemcpyHostToDevice, st));
    if (batch_keep && n_batches) HIPCHK(h, hipMemcpyAsync(h->d_gen_keep.p, batch_keep, sizeof(unsigned int) * (size_t)n_batches, hipMemcpyHostToDevice, st));
    HIPCHK(h, hipStreamSynchronize(st));  // (borrowed inputs)
    // TWO stream buffers: the deliveries of tile t + 1 are made (on a stream of their own) while tile t is tallied.  The two kernels
    // want different things of a CU -- the generator is arithmetic (the permutation, the scan over the batch lengths) and holds no
    // LDS, the tally of a packed round sits on its LDS with one wave per SIMD -- so they share the CUs instead of taking turns.
    // (Testing knob bit 21: one buffer, one stream, as before.)
    const bool overlap = (h->force_exact & 2097152) == 0 && R > T;
    const size_t buf_bytes = (((size_t)T * (size_t)A * stride + 64) + 255) & ~(size_t)255;
    HIPCHK(h, h->d_records_own.ensure(buf_bytes * (overlap ? 2 : 1)));
    HIPCHK(h, h->d_rec_off_own.ensure((size_t)T + 1));
    if (overlap && !h->stream_gen) {
        HIPCHK(h, hipStreamCreateWithFlags(&h->stream_gen, hipStreamNonBlocking));
        for (int i = 0; i < 2; ++i) {
            HIPCHK(h, hipEventCreateWithFlags(&h->ev_gen_ready[i], hipEventDisableTiming));
            HIPCHK(h, hipEventCreateWithFlags(&h->ev_gen_free[i], hipEventDisableTiming));
        }
        HIPCHK(h, hipEventCreateWithFlags(&h->ev_gen_inputs, hipEventDisableTiming));
    }
    hipLaunchKernelGGL(rapid::gen_offsets_kernel, dim3(grid_for(T + 1, 256)), dim3(256), 0, st, h->d_rec_off_own.p, (int)T, A);  // (the same for every tile)
    h->d_records = h->d_records_own.p;
    h->records_bytes = (unsigned long long)T * (unsigned long long)A * stride;
    h->d_rec_off = h->d_rec_off_own.p;
    h->offsets_on_device_only = false;
    h->rec_fmt = boundary ? rapid::kFmtBoundary : rapid::kFmtResident;
    h->n_receivers = (int)T;  // (what the launch geometry is chosen for)
    h->n_records_total = T * A;
    h->n_alert_set = A;
    h->d_alerts = h->d_alert_set.p;
    h->gen_cfg_id = h->config_id;
    h->gen_clean = clean && batch_keep == nullptr;
    // the deliveries are made here, by the library, from the declared alerts: it can vouch for them being copies itself
    h->trust_copies = true;
    h->streams_generated = true;
    HIPCHK(h, h->d_errflags.ensure(2));
    HIPCHK(h, h->d_stats.ensure(stats_words(h)));
    const size_t res_words = 10, ref_len = (size_t)h->max_cut + 1;
    const size_t back_bytes = res_words * 8 + ref_len * sizeof(int), seg_words = (back_bytes + 7) / 8;
    HIPCHK(h, h->d_voteback.ensure(seg_words));
    if ((rc = build_round_index(h))) return rc;  // once per round: every tile is a launch over the same alert set (also zeroes flags / statistics)
    if ((rc = ensure_tally_attrs(h))) return rc;
    if (!boundary && A > 0) {
        HIPCHK(h, h->d_gen_res.ensure((size_t)A));
        hipLaunchKernelGGL(rapid::gen_resolve_alerts_kernel, dim3(grid_for(A, 256)), dim3(256), 0, st, h->d_alert_set.p, A, (long long)h->config_id,
                           (unsigned int)h->n_nodes, h->d_entries.p, h->d_gen_res.p);
    }
    HIPCHK(h, h->d_gen_bat.ensure((size_t)std::max(n_batches, 1)));
    hipLaunchKernelGGL(rapid::gen_pack_batches_kernel, dim3(grid_for(n_batches, 256)), dim3(256), 0, st, h->d_gen_boff.p, n_batches,
                       boundary ? (const uint2*)nullptr : h->d_gen_res.p, batch_keep ? h->d_gen_keep.p : (const unsigned int*)nullptr, h->d_gen_bat.p);
    // per-receiver results for the whole population; node lists and bitmaps for one tile
    const size_t Rz = (size_t)std::max(R, 1);
    HIPCHK(h, h->d_emit.ensure(Rz));
    HIPCHK(h, h->d_nprop.ensure(Rz));
    HIPCHK(h, h->d_pcount.ensure(Rz));
    HIPCHK(h, h->d_fp.ensure(Rz));
    HIPCHK(h, h->d_props.ensure((size_t)T * (size_t)h->max_cut));
    HIPCHK(h, h->d_next.ensure(4));
    const int words = std::max((h->n_hot + 63) / 64, 1);
    HIPCHK(h, h->d_bitmaps.ensure((size_t)T * (size_t)words));
    const size_t acc_words = (size_t)rapid::kVoteAccWords + (size_t)words + ((size_t)h->max_cut + 1) / 2 + 1;
    HIPCHK(h, h->d_vacc.ensure(acc_words));
    unsigned long long* const acc = h->d_vacc.p;
    unsigned long long* const acc_bits = acc + rapid::kVoteAccWords;
    int* const acc_list = reinterpret_cast<int*>(acc_bits + words);
    unsigned long long* const d_res = h->d_voteback.p;  // (the tally's per-launch vote statistics land here too, and are consumed tile by tile)
    HIPCHK(h, h->d_gather.ensure(seg_words * (size_t)std::max(h->n_ranks, 1) + seg_words));
    unsigned long long* const d_block = h->d_gather.p + seg_words * (size_t)std::max(h->n_ranks, 1);  // this rank's answer block
    HIPCHK(h, h->d_hist.ensure((size_t)rapid::kVoteBuckets + 2));
    HIPCHK(h, h->d_mm.ensure(8));
    HIPCHK(h, h->d_winner.ensure(4));
    begin_round_result(h, out);
    unsigned long long* const hres = h->h_pinned;
    const unsigned long long my_tag = ~(unsigned long long)h->rank;

    unsigned long long target = 0ull;  // pass 0: the first voter's proposal is the candidate; a counting pass: the proposal with this fingerprint
    int passes = 0, tiles = 0;
    int verdict = RAPID_OK;
    for (int pass = 0; pass < 3; ++pass) {
        ++passes;
        HIPCHK(h, hipMemsetAsync(acc, 0, acc_words * 8, st));
        if (target != 0ull) HIPCHK(h, hipMemcpyAsync(acc + 6, &target, 8, hipMemcpyHostToDevice, st));  // (pageable, 8 bytes: staged by the runtime)
        // deliveries of the tile at `base` into stream buffer `buf`, on stream `gs`
        auto generate = [&](long long base, int buf, hipStream_t gs) {
            const int n = (int)std::min<long long>(T, R - base);
            hipLaunchKernelGGL(rapid::gen_streams_kernel, dim3(grid_for(n, rapid::kGenWavesPerBlock)), dim3(rapid::kGenWavesPerBlock * 64), 0, gs, h->d_gen_res.p,
                               h->d_alert_set.p, h->d_gen_bat.p, n_batches, batch_keep ? h->d_gen_keep.p : (const unsigned int*)nullptr, h->d_gen_rx.p + base,
                               n, A, (unsigned long long)seed, h->d_records_own.p + (size_t)buf * buf_bytes, boundary ? 1 : 0);
        };
        int buf = 0;
        if (overlap) {  // (the generator's inputs were written on the round's stream)
            HIPCHK_SYNC_GEN(hipEventRecord(h->ev_gen_inputs, st));
            HIPCHK_SYNC_GEN(hipStreamWaitEvent(h->stream_gen, h->ev_gen_inputs, 0));
            generate(0, 0, h->stream_gen);
            HIPCHK_SYNC_GEN(hipEventRecord(h->ev_gen_ready[0], h->stream_gen));
        }
        for (long long base = 0; base < R; base += T, buf ^= overlap ? 1 : 0) {
            const int n = (int)std::min<long long>(T, R - base);
            ++tiles;
            if (overlap) {
                if (base + T < R) {  // the next tile's deliveries, into the buffer the tile before this one was tallied from
                    if (base > 0) HIPCHK_SYNC_GEN(hipStreamWaitEvent(h->stream_gen, h->ev_gen_free[buf ^ 1], 0));
                    generate(base + T, buf ^ 1, h->stream_gen);
                    HIPCHK_SYNC_GEN(hipEventRecord(h->ev_gen_ready[buf ^ 1], h->stream_gen));
                }
                HIPCHK_SYNC_GEN(hipStreamWaitEvent(st, h->ev_gen_ready[buf], 0));
            } else {
                generate(base, 0, st);
            }
            h->d_records = h->d_records_own.p + (size_t)buf * buf_bytes;
            h->n_receivers = n;
            h->out_base = base;
            if ((rc = launch_tally(h))) {
                if (overlap) (void)hipStreamSynchronize(h->stream_gen);
                return rc;
            }
            if (overlap) HIPCHK_SYNC_GEN(hipEventRecord(h->ev_gen_free[buf], st));
            hipLaunchKernelGGL(rapid::vote_acc_pick_kernel, dim3(1), dim3(1024), 0, st, d_res, h->d_fp.p + base, h->d_pcount.p + base, h->d_props.p, h->max_cut,
                               h->d_bitmaps.p, words, n, h->d_errflags.p, acc, acc_bits, acc_list);
            hipLaunchKernelGGL(rapid::vote_acc_count_kernel, dim3(std::max(1u, grid_for((long long)n * 64, 1024))), dim3(1024), 0, st, h->d_fp.p + base,
                               h->d_pcount.p + base, h->d_bitmaps.p, words, n, acc, acc_bits);
            h->tiled_last_base = (int)base;
            h->tiled_last_n = n;
        }
        h->tally_votes_valid = false;
        h->tally_settled_valid = false;
        hipLaunchKernelGGL(rapid::vote_acc_finish_kernel, dim3(8), dim3(256), 0, st, acc, acc_list, h->max_cut, d_block, reinterpret_cast<int*>(d_block + res_words));
        HIPCHK(h, hipGetLastError());
        const unsigned long long* ans = nullptr;
        bool settled = false;
        if (h->comm) {  // the round's one collective: every rank's block to everybody, merged identically everywhere
            NCCLCHK(h, ncclAllGather(d_block, h->d_gather.p, seg_words, ncclUint64, h->comm, st));
            hipLaunchKernelGGL(rapid::vote_merge_kernel, dim3(1), dim3(256), 0, st, h->d_gather.p, h->n_ranks, (int)seg_words, (int)res_words, h->max_cut,
                               (long long)out->quorum, reinterpret_cast<volatile unsigned long long*>(h->d_mail + 64),
                               reinterpret_cast<volatile unsigned int*>(h->d_mail) + 14, ++h->mail_seq);
            HIPCHK(h, hipGetLastError());
            if ((rc = await_mail(h, 14, h->mail_seq))) return rc;
            ans = reinterpret_cast<const unsigned long long*>(h->h_mail + 64);
            settled = ans[9] == 1ull;
        } else {
            HIPCHK(h, hipMemcpyAsync(hres, d_block, back_bytes, hipMemcpyDeviceToHost, st));
            HIPCHK(h, hipStreamSynchronize(st));
            ans = hres;
            const unsigned long long votes = ans[1], voters = ans[2];
            settled = voters == 0ull || votes == voters || (long long)votes >= (long long)out->quorum || ans[6] != 0ull;
        }
        if (settled) {
            verdict = decode_vote_answer(h, ans, reinterpret_cast<const int*>(ans + res_words), out);
            break;
        }
        // The candidate has no quorum and the voters are not unanimous: the exact plurality is owed (R/FastPaxos.java:141-150 counts
        // every distinct proposal).  Every receiver's fingerprint is still here: their positional histogram (summed over the ranks)
        // says how many votes the most popular proposal can have at most; below the quorum nothing is decided, whatever it is.
        // Otherwise the tiles are taken once more with THAT proposal -- the one holding the winning bucket's fingerprint -- as the
        // candidate, its voters verified bit for bit like the first candidate's.
        unsigned long long win[4] = {0, 0, 0, 0}, mm[2] = {0, 0};
        bool found = false;
        for (unsigned long long salt = 0; salt < 4 && !found; ++salt) {
            HIPCHK(h, hipMemsetAsync(h->d_hist.p, 0, ((size_t)rapid::kVoteBuckets + 2) * 8, st));
            HIPCHK(h, hipMemsetAsync(h->d_mm.p, 0, 64, st));
            if (R) hipLaunchKernelGGL(rapid::vote_histogram_kernel, dim3(grid_for(R, 256)), dim3(256), 0, st, h->d_fp.p, h->d_pcount.p, R, salt, h->d_hist.p);
            if (h->comm) NCCLCHK(h, ncclAllReduce(h->d_hist.p, h->d_hist.p, (size_t)rapid::kVoteBuckets + 2, ncclUint64, ncclSum, h->comm, st));
            hipLaunchKernelGGL(rapid::vote_winner_kernel, dim3(1), dim3(1024), 0, st, h->d_hist.p, h->d_winner.p);
            if (R) hipLaunchKernelGGL(rapid::vote_bucket_minmax_kernel, dim3(grid_for(R, 256)), dim3(256), 0, st, h->d_fp.p, h->d_pcount.p, R, salt, h->d_winner.p, h->d_mm.p, my_tag);
            if (h->comm) NCCLCHK(h, ncclAllReduce(h->d_mm.p, h->d_mm.p, 2, ncclUint64, ncclMax, h->comm, st));
            HIPCHK(h, hipMemcpyAsync(win, h->d_winner.p, sizeof win, hipMemcpyDeviceToHost, st));
            HIPCHK(h, hipMemcpyAsync(mm, h->d_mm.p, sizeof mm, hipMemcpyDeviceToHost, st));
            HIPCHK(h, hipStreamSynchronize(st));
            out->votes_total = (int64_t)win[2];
            out->votes_winner = (int64_t)win[1];
            out->distinct_local = (int32_t)std::min<unsigned long long>(win[3], 0x7FFFFFFFull);
            if ((long long)win[1] < (long long)out->quorum) {  // no proposal can have a quorum
                found = true;
                target = 0ull;
            } else if (mm[0] == ~mm[1]) {  // the winning bucket holds one fingerprint: the proposal to count
                found = true;
                target = mm[0];
            }
        }
        if (!found) return fail(h, RAPID_ECOLLISION, "winning vote bucket stayed impure under 4 salts");
        if (target == 0ull || pass == 2) break;  // undecided (pass 2 cannot get here with a quorum left uncounted)
    }
    h->out_base = 0;
    h->n_receivers = R;
    h->tiled_total = R;
    h->tiled_block = d_block;
    h->tallied = true;
    h->tiled_ms[0] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    h->tiled_ms[1] = tiles;
    h->tiled_ms[2] = passes;
    h->tiled_ms[3] = (double)R * (double)A * passes / 1e6;
    return verdict == kVoteNextSalt ? fail(h, RAPID_ECOLLISION, "the accumulated answer is impure") : verdict;
}

int rapid_sim_round_tiled_info(rapid_engine* h, double out[4]) {
    if (!h || !out) return RAPID_EINVAL;
    for (int i = 0; i < 4; ++i) out[i] = h->tiled_ms[i];
    return RAPID_OK;
}

int rapid_apply_cut(rapid_engine* h, const int32_t* cut, int32_t n, int64_t* new_config_id) {
    if (!h || n < 0 || (n > 0 && !cut)) return RAPID_EINVAL;
    if (!h->view_built) return fail(h, RAPID_ESTATE, "view not built");
    int rc = use_device(h);
    if (rc) return rc;
    // A failing call leaves the membership untouched: the flags are set / cleared in place -- the cost of the call is the cut's, not
    // the registry's -- and put back if the cut does not validate or the device fails on the way (view_change_failed)
    for (int i = 0; i < n; ++i)
        if (cut[i] < 0 || cut[i] >= h->n_nodes) return fail(h, RAPID_EINVAL, "node index %d out of range", cut[i]);
    const bool timing = env_knob("RAPID_TIME_VIEW") != nullptr;
    auto tp = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!timing) return;
        const auto t = std::chrono::steady_clock::now();
        std::fprintf(stderr, "apply_cut %-32s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t - tp).count());
        tp = t;
    };
    std::vector<std::pair<int64_t, int64_t>> added;
    for (int i = 0; i < n; ++i) {
        const int node = cut[i];
        if (h->member[(size_t)node]) {
            h->member[(size_t)node] = 0;  // ringDelete (R/MembershipService.java:399-400)
        } else {
            added.push_back({h->id_hi[(size_t)node], h->id_lo[(size_t)node]});
            h->member[(size_t)node] = 1;  // ringAdd (R/MembershipService.java:404-407)
        }
    }
    auto undo = [&]() {  // (in reverse: a node named twice goes back through both of its states)
        for (int i = n - 1; i >= 0; --i) h->member[(size_t)cut[i]] ^= 1;
    };
    if (!added.empty()) {  // ringAdd :127-129: an identifier seen before -- in an earlier configuration, or earlier in this cut
        lap("flags");
        rapid::sort_node_ids(added);
        bool seen = std::adjacent_find(added.begin(), added.end()) != added.end();
        lap("joiners' identifiers sorted");
        if (!seen && (rc = ids_seen_any(h, added, &seen))) {
            undo();
            return rc;
        }
        if (seen) {
            undo();
            return fail(h, RAPID_EUUID_SEEN, "the identifier of a joiner in the cut was already seen");
        }
    }
    lap("identifiers seen before?");
    const int n_ids_before = h->n_ids_dev;
    const size_t pending_before = h->ids_pending.size();
    h->changed.insert(h->changed.end(), cut, cut + n);
    h->ids_pending.insert(h->ids_pending.end(), added.begin(), added.end());
    rc = rebuild_view(h);  // new rings, tables, configuration id; cutDetection.clear() == fresh state next tally
    if (rc) {  // (a device error on the way: the membership goes back to what it was and nothing is built on half-changed rings)
        undo();
        if (h->ids_pending.size() > pending_before) h->ids_pending.resize(pending_before);
        view_change_failed(h, n_ids_before);
        return rc;
    }
    lap("rebuild_view");
    if (new_config_id) *new_config_id = h->config_id;
    return RAPID_OK;
}

int rapid_sim_round(rapid_engine* h, int32_t apply, rapid_round_result* out, int64_t* new_config_id) {
    int rc = rapid_sim_tally(h);
    if (rc) return rc;
    if ((rc = rapid_sim_count_votes(h, out))) return rc;
    if (out->decided && apply) {
        const std::vector<int> cut = h->decided_cut;
        if ((rc = rapid_apply_cut(h, cut.data(), (int)cut.size(), new_config_id))) return rc;
        h->decided_cut = cut;
        h->have_decision = true;
    } else if (new_config_id) {
        *new_config_id = h->config_id;
    }
    return RAPID_OK;
}

// A round whose deliveries and distinct alerts already lie in device memory, in ONE call: attach in place, declare in place, the
// trust level, index + tally + vote count (+ apply).  What rapid_sim_attach_streams_device + rapid_sim_set_alert_set_device +
// rapid_sim_trust_alert_copies + rapid_sim_round do one after the other -- the same checks, the same state afterwards -- without a
// host paying four or five crossings of its foreign-function boundary per round (a JNI crossing with its pinning of arguments
// costs about what a kernel launch does; from Python it is ~1.5 us per call: 5 us of a 450 us round at N = 10^4).
int rapid_sim_round_device(rapid_engine* h, const void* d_records, uint64_t records_bytes, const int64_t* d_rec_off, int32_t n_receivers,
                           const void* d_alerts, uint64_t alerts_bytes, int64_t n_alerts, int32_t trust, int32_t apply,
                           rapid_round_result* out, int64_t* new_config_id) {
    if (!h || !out) return RAPID_EINVAL;
    if (trust < 0 || trust > 2) return fail(h, RAPID_EINVAL, "trust level %d (0, 1 or 2)", trust);
    int rc = rapid_sim_attach_streams_device(h, d_records, records_bytes, d_rec_off, n_receivers);
    if (rc) return rc;
    if (d_alerts != nullptr || n_alerts > 0) {
        if ((rc = rapid_sim_set_alert_set_device(h, d_alerts, alerts_bytes, n_alerts))) return rc;
    }
    if ((rc = rapid_sim_trust_alert_copies(h, trust))) return rc;
    return rapid_sim_round(h, apply, out, new_config_id);
}

// ------------------------------------------------------------------------------------------- multi-GPU
int rapid_comm_unique_id(uint8_t out[RAPID_UNIQUE_ID_BYTES]) {
    static_assert(sizeof(ncclUniqueId) <= RAPID_UNIQUE_ID_BYTES, "unique id does not fit");
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return RAPID_EDEVICE;
    std::memset(out, 0, RAPID_UNIQUE_ID_BYTES);
    std::memcpy(out, &id, sizeof id);
    return RAPID_OK;
}

int rapid_engine_comm_init(rapid_engine* h, const uint8_t id_bytes[RAPID_UNIQUE_ID_BYTES], int32_t rank, int32_t n_ranks) {
    if (!h || !id_bytes || n_ranks <= 0 || rank < 0 || rank >= n_ranks) return RAPID_EINVAL;
    int rc = use_device(h);
    if (rc) return rc;
    ncclUniqueId id;
    std::memcpy(&id, id_bytes, sizeof id);
    NCCLCHK(h, ncclCommInitRank(&h->comm, n_ranks, id, rank));
    h->rank = rank;
    h->n_ranks = n_ranks;
    return RAPID_OK;
}

int rapid_engine_comm_info(rapid_engine* h, int32_t* rank, int32_t* n_ranks) {
    if (!h || !rank || !n_ranks) return RAPID_EINVAL;
    *rank = 0;
    *n_ranks = 1;
    if (!h->comm) return RAPID_OK;  // no communicator: a population held by one engine
    int r = 0, n = 0;
    NCCLCHK(h, ncclCommUserRank(h->comm, &r));
    NCCLCHK(h, ncclCommCount(h->comm, &n));  // what the communicator itself says, not what the caller asked for
    *rank = r;
    *n_ranks = n;
    return RAPID_OK;
}

// --------------------------------------------------------------------------------------- instrumentation
void* rapid_engine_stream(rapid_engine* h) { return h ? (void*)h->stream : nullptr; }

int rapid_engine_sync(rapid_engine* h) {
    if (!h) return RAPID_EINVAL;
    int rc = use_device(h);
    if (rc) return rc;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    HIPCHK(h, hipGetLastError());
    return RAPID_OK;
}

int rapid_sim_stats(rapid_engine* h, uint64_t stats[8]) {
    if (!h || !stats) return RAPID_EINVAL;
    if (!h->tallied) return fail(h, RAPID_ESTATE, "no tally has run");
    int rc = use_device(h);
    if (rc) return rc;
    // one row of eight counters per workgroup (no contended atomics in the kernel): summed here
    const size_t rows = (size_t)std::max(h->grid_blocks, 1);
    std::vector<uint64_t> per_block(rows * 8);
    HIPCHK(h, hipMemcpyAsync(per_block.data(), h->d_stats.p, rows * 64, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    for (int i = 0; i < 8; ++i) stats[i] = 0;
    for (size_t b = 0; b < rows; ++b)
        for (int i = 0; i < 8; ++i) stats[i] += per_block[b * 8 + (size_t)i];
    return RAPID_OK;
}

int rapid_sim_time_tally(rapid_engine* h, int32_t reps, float* ms_avg) {
    if (!h || reps <= 0 || !ms_avg) return RAPID_EINVAL;
    int rc = use_device(h);
    if (rc) return rc;
    if ((rc = prepare_tally(h))) return rc;
    if (h->n_receivers == 0) return fail(h, RAPID_ESTATE, "no receivers");
    if (!h->ev0) HIPCHK(h, hipEventCreate(&h->ev0));
    if (!h->ev1) HIPCHK(h, hipEventCreate(&h->ev1));
    const hipEvent_t e0 = h->ev0, e1 = h->ev1;
    HIPCHK(h, hipMemsetAsync(h->d_stats.p, 0, (size_t)64 * ((size_t)std::max(h->grid_blocks, 1) + 1), h->stream));
    launch_tally(h);  // untimed warm-up
    HIPCHK(h, hipEventRecord(e0, h->stream));
    for (int i = 0; i < reps; ++i) launch_tally(h);
    HIPCHK(h, hipEventRecord(e1, h->stream));
    HIPCHK(h, hipEventSynchronize(e1));
    HIPCHK(h, hipGetLastError());
    float ms = 0.f;
    HIPCHK(h, hipEventElapsedTime(&ms, e0, e1));
    *ms_avg = ms / (float)reps;
    h->tallied = true;
    return RAPID_OK;
}

#ifdef RAPID_TEST_BUILD
// Measurement probe: streams the loaded records with the tally kernel's access pattern and no processing.
// variant: 0 = 2 KiB tiles x 8 in flight, 1 = 4 KiB x 4, 2 = 8 KiB x 2, 3 = 2 KiB x 4; waves = waves per block.
int rapid_debug_stream_probe(rapid_engine* h, int32_t variant, int32_t waves, int32_t reps, float* ms_avg) {
    if (!h || reps <= 0 || !ms_avg || waves <= 0 || waves > 16) return RAPID_EINVAL;
    int rc = use_device(h);
    if (rc) return rc;
    if (!h->streams_loaded || h->n_receivers == 0) return fail(h, RAPID_ESTATE, "no alert streams loaded");
    HIPCHK(h, h->d_next.ensure(4));
    if (!h->ev0) HIPCHK(h, hipEventCreate(&h->ev0));
    if (!h->ev1) HIPCHK(h, hipEventCreate(&h->ev1));
    const hipEvent_t e0 = h->ev0, e1 = h->ev1;
    int per_cu = 16;  // waves per CU
    if (const char* e = env_knob("RAPID_PROBE_WAVES_PER_CU")) per_cu = std::max(1, std::min(32, atoi(e)));
    const dim3 grid((unsigned)h->num_cus * (unsigned)std::max(1, per_cu / waves)), block((unsigned)waves * 64u);
    size_t lds_pad = 0;
    if (const char* e = env_knob("RAPID_PROBE_LDS_PAD")) lds_pad = (size_t)atoi(e);
    if (lds_pad > 0)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(rapid::dma_probe_kernel<6>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    unsigned int ring_off = 0;
    if (const char* e = env_knob("RAPID_PROBE_RING_OFFSET")) ring_off = (unsigned int)atoi(e);
    const unsigned long long rec_b = h->rec_fmt == rapid::kFmtBoundary ? 20ull : 8ull;
    auto launch = [&]() {
        (void)hipMemsetAsync(h->d_next.p, 0, 4, h->stream);
        (void)hipMemcpyAsync(h->d_next.p + 2, &ring_off, 4, hipMemcpyHostToDevice, h->stream);
        switch (variant) {
            case 0: hipLaunchKernelGGL((rapid::stream_probe_kernel<2, 8>), grid, block, 0, h->stream, const_cast<unsigned char*>(h->d_records), h->records_bytes, h->d_rec_off, h->n_receivers, h->d_next.p, h->d_next.p + 1, rec_b); break;
            case 1: hipLaunchKernelGGL((rapid::stream_probe_kernel<4, 4>), grid, block, 0, h->stream, const_cast<unsigned char*>(h->d_records), h->records_bytes, h->d_rec_off, h->n_receivers, h->d_next.p, h->d_next.p + 1, rec_b); break;
            case 2: hipLaunchKernelGGL((rapid::stream_probe_kernel<8, 2>), grid, block, 0, h->stream, const_cast<unsigned char*>(h->d_records), h->records_bytes, h->d_rec_off, h->n_receivers, h->d_next.p, h->d_next.p + 1, rec_b); break;
            case 9: hipLaunchKernelGGL((rapid::dma_probe_kernel<6>), grid, block, (size_t)waves * 6 * 1024 + lds_pad, h->stream, const_cast<unsigned char*>(h->d_records), h->records_bytes, h->d_rec_off, h->n_receivers, h->d_next.p, h->d_next.p + 1, rec_b); break;
            case 7: hipLaunchKernelGGL((rapid::dma_probe_kernel<4>), grid, block, (size_t)waves * 4 * 1024, h->stream, const_cast<unsigned char*>(h->d_records), h->records_bytes, h->d_rec_off, h->n_receivers, h->d_next.p, h->d_next.p + 1, rec_b); break;
            case 8: hipLaunchKernelGGL((rapid::dma_probe_kernel<8>), grid, block, (size_t)waves * 8 * 1024, h->stream, const_cast<unsigned char*>(h->d_records), h->records_bytes, h->d_rec_off, h->n_receivers, h->d_next.p, h->d_next.p + 1, rec_b); break;
            case 4: hipLaunchKernelGGL((rapid::stream_probe_kernel<1, 8>), grid, block, 0, h->stream, const_cast<unsigned char*>(h->d_records), h->records_bytes, h->d_rec_off, h->n_receivers, h->d_next.p, h->d_next.p + 1, rec_b); break;
            case 5: hipLaunchKernelGGL((rapid::stream_probe_kernel<1, 16>), grid, block, 0, h->stream, const_cast<unsigned char*>(h->d_records), h->records_bytes, h->d_rec_off, h->n_receivers, h->d_next.p, h->d_next.p + 1, rec_b); break;
            case 6: hipLaunchKernelGGL((rapid::stream_probe_kernel<1, 4>), grid, block, 0, h->stream, const_cast<unsigned char*>(h->d_records), h->records_bytes, h->d_rec_off, h->n_receivers, h->d_next.p, h->d_next.p + 1, rec_b); break;
            default: hipLaunchKernelGGL((rapid::stream_probe_kernel<2, 4>), grid, block, 0, h->stream, const_cast<unsigned char*>(h->d_records), h->records_bytes, h->d_rec_off, h->n_receivers, h->d_next.p, h->d_next.p + 1, rec_b); break;
        }
    };
    launch();
    HIPCHK(h, hipEventRecord(e0, h->stream));
    for (int i = 0; i < reps; ++i) launch();
    HIPCHK(h, hipEventRecord(e1, h->stream));
    HIPCHK(h, hipEventSynchronize(e1));
    HIPCHK(h, hipGetLastError());
    float ms = 0.f;
    HIPCHK(h, hipEventElapsedTime(&ms, e0, e1));
    *ms_avg = ms / (float)reps;
    return RAPID_OK;
}

#endif  // RAPID_TEST_BUILD

int rapid_sim_pass_times(rapid_engine* h, float out[4]) {
    if (!h || !out) return RAPID_EINVAL;
    int rc = use_device(h);
    if (rc) return rc;
    if (h->index_ms_pending && h->ev_idx0 && h->ev_idx1) {
        h->index_ms_pending = false;
        if (hipEventSynchronize(h->ev_idx1) == hipSuccess) (void)hipEventElapsedTime(&h->index_ms, h->ev_idx0, h->ev_idx1);
        (void)hipGetLastError();
    }
    out[0] = h->index_ms;
    out[1] = 0.f;  // (there is no resolve pass any more: a delivered record is read once, by the tally)
    out[2] = h->generate_ms;
    out[3] = 0.f;
    return RAPID_OK;
}

int rapid_sim_index_info(rapid_engine* h, int32_t info[8], float* index_ms) {
    if (!h || !info) return RAPID_EINVAL;
    int rc = use_device(h);
    if (rc) return rc;
    if (index_ms) h->time_index = true;  // (the next build is timed: this call's own if the index is stale)
    if ((rc = prepare_tally(h))) return rc;  // builds the index if it is stale
    info[0] = h->n_hot;
    info[1] = h->n_adj;
    info[2] = h->waves_per_block;
    info[3] = h->grid_blocks;
    info[4] = h->lds_bytes;
    info[5] = tally_is_trusted(h) ? 1 : 0;
    info[6] = h->dict_mode;  // 3 = resolved records (no lookup in the tally); 0 / 1 / 2 = tables in memory / direct in LDS / compressed in LDS
    // (bit 2: the tally runs the instantiation that leaves the records' configuration ids in their cache lines -- launch_tally)
    const bool ids_skipped = records_known_current(h) &&
                             (h->dict_mode == rapid::kDictMemory || (!h->packed && (h->dict_mode == rapid::kDictDirect || h->dict_mode == rapid::kDictCompressed)));
    info[7] = (h->n_alert_set >= 0 ? 1 : 0) | (h->q4_live ? 2 : 0) | (ids_skipped ? 4 : 0);
    if (h->index_ms_pending && h->ev_idx0 && h->ev_idx1) {
        h->index_ms_pending = false;
        if (hipEventSynchronize(h->ev_idx1) == hipSuccess) (void)hipEventElapsedTime(&h->index_ms, h->ev_idx0, h->ev_idx1);
        (void)hipGetLastError();
    }
    if (index_ms) *index_ms = h->index_ms;
    return RAPID_OK;
}

#ifdef RAPID_TEST_BUILD
namespace rapid {
__global__ void debug_wild_store_kernel(unsigned long long* p) { p[threadIdx.x] = 0xFA17ull; }
}  // namespace rapid
// (see the header: the test suite's fault tolerance, exercised with a real fault when somebody asks for it)
int rapid_debug_device_fault(rapid_engine* h) {
    if (!h) return RAPID_EINVAL;
    int rc = use_device(h);
    if (rc) return rc;
    hipLaunchKernelGGL(rapid::debug_wild_store_kernel, dim3(1), dim3(64), 0, h->stream, reinterpret_cast<unsigned long long*>(0x10ull));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return RAPID_OK;  // (not reached on a device that faults)
}

int rapid_debug_block_stats(rapid_engine* h, uint64_t* out, int32_t cap_rows, int32_t* rows_out) {
    if (!h || !rows_out || cap_rows < 0 || (cap_rows > 0 && !out)) return RAPID_EINVAL;
    int rc = use_device(h);
    if (rc) return rc;
    const int rows = std::max(h->grid_blocks, 1);
    *rows_out = rows;
    if (rows > cap_rows) return RAPID_ECAPACITY;
    HIPCHK(h, hipMemcpyAsync(out, h->d_stats.p, (size_t)rows * 64, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return RAPID_OK;
}

#endif  // RAPID_TEST_BUILD

int rapid_sim_set_force_exact(rapid_engine* h, int32_t on) {
    if (!h) return RAPID_EINVAL;
    if (((h->force_exact ^ on) & (64 | 128 | 256 | 4096 | 8192 | 1048576)) != 0) h->index_valid = false;  // the launch geometry depends on the instantiation and on where the dictionary lives
    h->force_exact = on;
    return RAPID_OK;
}


}  // extern "C"
