// Alert-tally kernels for gfx950 (CDNA4, wave64) -- the hot path.
//
// What they compute (reference semantics, R/ = /root/reference/rapid/src/main/java/com/vrg/rapid/):
//   per receiver: MembershipService.handleMessage(BatchedAlertMessage)  R/MembershipService.java:300-354
//                 + filterAlertMessages                                 R/MembershipService.java:644-675
//                 + MultiNodeCutDetector.aggregateForProposal           R/MultiNodeCutDetector.java:76-128
//                 + MultiNodeCutDetector.invalidateFailingEdges         R/MultiNodeCutDetector.java:137-164
//
// Mapping onto the machine:
//   * one wavefront per simulated receiver, persistent; receivers are dealt round-robin (wave-major); a workgroup is
//     W such wavefronts (one workgroup per CU, W chosen by the host) that share read-only per-round tables in LDS;
//   * the receiver's whole detector state is one word per SLOT in LDS -- a slot is a "hot" subject: one the round's
//     alert set names on >= L distinct rings, the only kind that can ever reach the L watermark at any receiver
//     (index_kernels.h builds the node->slot dictionary once per loaded stream set; reports about other subjects can
//     never change any receiver's outcome and only contribute to seenLinkDownEvents): bits 0..K-1 = rings reported,
//     bit 14 = already flushed into an emitted proposal;
//   * the delivered stream is read ONCE from HBM and never passes through registers: LDS-DMA loads (lds_dma.h), 1 KiB
//     per wave instruction, kDepth of them in flight per wave, land in the wave's own 10 KiB LDS ring; completion is
//     counted by hand (s_waitcnt vmcnt(kDepth - 1) = the oldest has landed).  The tally loop touches global memory
//     for nothing else;
//   * LEAN path per window of up to 192 records (three per lane): order-free ds_or_rtn on the masks, committed under a
//     certificate that the reference cannot emit inside the window (a witness slot that provably stays in
//     preProposal; or no possible H crossing; or fewer H crossings than updatesInProgress); the implicit edge
//     invalidation is deferred, incremental (only the slots that crossed L) and walks only the round's "hot"
//     adjacency (pairs whose both ends can reach L at all);
//   * without a certificate the window is rolled back (each lane clears exactly the bits it set) and taken by the
//     CAREFUL path (<= 64 records, invalidation applied immediately, exact crossing count, halving) and finally by the
//     EXACT path, record by record;
//   * after the batch that announces a proposal the receiver ignores the rest of its stream
//     (announcedProposal, R/MembershipService.java:318-319) -- the wave stops reading it.
//
// No MFMA: the path is integer scatter/popcount.  All LDS is carved from the 16-byte aligned dynamic segment.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include <lds_dma.h>

namespace rapid {

constexpr int kWave = 64;
constexpr int kRecBytes = 20;
constexpr int kSlotBytes = 1024;                     // one LDS-DMA wave instruction: 64 lanes x 16 B
#ifndef RAPID_QUARTERS
#define RAPID_QUARTERS 3
#endif
constexpr int kRepickAfter = 8 * 192;                // RAPID_FAST_WINDOW: records after which the first witness is reconsidered
constexpr int kEndGameRunning = 16;                   // RAPID_CAREFUL_HINT: updatesInProgress at which the lean path is left for good
constexpr int kQuarters = RAPID_QUARTERS;            // records per lane in a lean window
constexpr int kLeanWindow = kQuarters * kWave;       // 192 records = 3840 B (2 and 4 per lane measured slower: profiles/)
constexpr int kWindowSlots = (kLeanWindow * kRecBytes + kSlotBytes - 1) / kSlotBytes + 1;  // slots a window can touch at any alignment
// RAPID_LEAN_V2 (default 0 = the kernel that was measured in round 1): the same lean window with fewer instructions --
// specialised on "a DOWN report has been seen" (no per-part test of it), lane masks built from the compares themselves
// instead of a ballot of their conjunction, the per-lane predicate (not a bit test of the ballot) guarding the atomics.
// Emulator-verified; to be timed against the default before it replaces it (scripts/build_variant.sh).
#ifndef RAPID_LEAN_V2
#define RAPID_LEAN_V2 0
#endif
// RAPID_CAREFUL_HINT (default 0): when the careful path cannot exclude an emission in a sub-chunk, retry with the prefix
// that holds fewer explicit H crossings than updatesInProgress (read off the crossing mask) instead of halving blindly,
// and go record by record at once when the first record is the critical one.  Only the size of the next attempt
// changes: every attempt is certified or replayed exactly as before.
#ifndef RAPID_CAREFUL_HINT
#define RAPID_CAREFUL_HINT 0
#endif
// RAPID_EARLY_CERT (default 0): a fourth certificate for the lean window while there is no witness yet (the first windows
// of a receiver): if no subject the window touches can reach H even when credited with every implicit report it can
// ever get -- popc(state | subject_mask) < H -- and the same held for every subject before the window, nothing crosses H
// in it, so nothing is emitted; any entrant of such a window is a witness for the next one.  Tables in LDS only.
#ifndef RAPID_EARLY_CERT
#define RAPID_EARLY_CERT 0
#endif
// RAPID_FAST_WINDOW (default 0): while a witness exists, full windows are applied WITHOUT looking at what the reports do
// to their subjects: the bits the window adds to the witness are worked out beforehand (its own state + the window's
// records about it), the window is applied only if the witness provably stays below H -- so nothing is ever rolled
// back -- and the atomics return nothing.  Which subjects crossed L meanwhile is not tracked at all: the owed implicit
// reports are applied by one pass over the round's (subject, observer, ring) pairs among the hot slots -- the
// reference's literal invalidateFailingEdges -- a few hundred pairs, 64 per step.
#ifndef RAPID_FAST_WINDOW
#define RAPID_FAST_WINDOW 0
#endif
// RAPID_DMA_PAIRS (default 0): the stream is topped up two KiB per loop trip (one wait for two landed KiB, two loads)
// instead of one -- half the scalar bookkeeping and branches per KiB.
#ifndef RAPID_DMA_PAIRS
#define RAPID_DMA_PAIRS 0
#endif
#ifndef RAPID_RING_SLOTS
#define RAPID_RING_SLOTS 10
#endif
constexpr int kDepth = RAPID_RING_SLOTS - kWindowSlots;  // KiB kept in flight per wave (the rest of the ring)
static_assert(kDepth >= 1, "ring too small for the window");
constexpr int kRingSlots = kWindowSlots + kDepth;    // LDS ring the windows are decoded from
constexpr int kRingBytes = kSlotBytes * kRingSlots;  // 10 KiB = 512 records exactly: records never straddle the ring's end
// With a ring of a whole number of records (10 KiB = 512) no record straddles its end and one wrap per lane is enough;
// otherwise every dword of a record is wrapped on its own (a smaller ring, more waves per CU, two more VALU per read).
constexpr bool kRingRecordAligned = kRingBytes % kRecBytes == 0;
constexpr int kPendCap = 128;                        // slots that crossed L and still await invalidation
constexpr int kUndoCap = 128;                        // implicit bits set inside one sub-chunk
constexpr int kMaxWavesPerBlock = 16;
constexpr uint32_t kFlushed = 1u << 14;

// dictionary entry (16 bit): bit 15 = node is a member, bit 14 = slot has hot adjacency, bits 0..13 = slot
constexpr unsigned int kDictMember = 1u << 15;
constexpr unsigned int kDictHasAdj = 1u << 14;
constexpr unsigned int kSlotMask = 0x3FFFu;
constexpr unsigned int kNoSlot = 0x3FFFu;            // at most 16382 subjects per round

// Per-round index over the loaded alert set (built by index_kernels.h; all device pointers).
// Slots [0, n_hot) are the "hot" subjects -- those named on >= L distinct rings by the round's alert set, the only
// ones that can ever enter preProposal/proposal at any receiver -- in ascending node order.
struct RoundIndex {
    const unsigned short* dict;     // [n_nodes] node -> kDictMember | kDictHasAdj | slot (kNoSlot: not hot)
    const int* node_of_slot;        // [n_hot]
    const unsigned short* adj_off;  // [n_hot + 1] CSR over hot slots
    const unsigned int* adj;        // [n_adj] other_slot | ring << 16 | role << 20 (role 1: `other` is the subject)
    int n_hot, n_adj;
};

struct TallyParams {
    const unsigned char* records;      // packed rapid_alert_record[]
    unsigned long long records_bytes;  // readable bytes at `records` (multiple of 16, covering the last record)
    const long long* rec_off;          // [R+1], in records
    int n_receivers;
    int n_nodes;
    int K, H, L;
    long long cfg_id;
    RoundIndex idx;
    int* emit_batch;                  // [R]
    int* num_proposals;               // [R]
    int* prop_count;                  // [R]; -1 if the proposal did not fit prop_cap
    unsigned long long* fingerprint;  // [R]
    int* props;                       // [R][prop_cap] ascending node index
    int prop_cap;
    unsigned long long* stats;        // [workgroups][8], accumulated over launches
    int waves_per_block;
    int flags;                        // bit0: exact path only, bit3: careful path only (both for tests); bit5: stream only
};

__host__ __device__ inline int align16(int x) { return (x + 15) & ~15; }
// LDS budget: shared tables (only when they are staged in LDS) + per-wave detector state, ring, lists
__host__ __device__ inline int tally_pairs_bytes(int n_adj) {  // RAPID_FAST_WINDOW: [count, (subject, observer, ring) ...]
    return RAPID_FAST_WINDOW ? align16((n_adj / 2 + 1) * 4) : 0;
}
__host__ __device__ inline int tally_shared_bytes(int n_nodes, int n_hot, int n_adj) {
    return align16(n_nodes * 2) + align16((n_hot + 1) * 2) + align16(n_adj * 4) + align16(n_hot * 4) + align16(n_hot * 2) +
           tally_pairs_bytes(n_adj);
}
// per-workgroup statistics accumulator at the very end of the dynamic LDS segment
constexpr int kBlockStatsBytes = 64;
__host__ __device__ inline int tally_wave_bytes(int n_slots) {
    return align16(n_slots * 4) + kRingBytes + align16(kPendCap * 2) + kUndoCap * 4;
}

// ---- small wave helpers ---------------------------------------------------------------------------------------
// A receiver is owned by ONE wavefront; that wave's LDS operations execute in program order, so cross-lane
// hand-offs through its private LDS region only need the compiler not to reorder or cache them -- no s_barrier
// and no s_waitcnt vmcnt (which would drain the tile prefetch that is deliberately left in flight).
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ unsigned long long lanes_lt(int lane) { return (1ull << lane) - 1ull; }
// number of set bits of m below this lane's own bit (v_mbcnt_lo / v_mbcnt_hi)
__device__ __forceinline__ int rank_below(unsigned long long m) {
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u));
}
// lane mask of a per-lane predicate, straight from the compare (no bool -> int -> compare round trip)
__device__ __forceinline__ unsigned long long wave_ballot(bool pred) { return __builtin_amdgcn_ballot_w64(pred); }
// Values that are the same in every lane are kept provably uniform (SGPRs, scalar branches): everything derived
// from them -- loop bounds, the detector's counters -- then costs scalar instead of exec-masked vector code.
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ unsigned int uniform(unsigned int v) { return (unsigned int)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ int lane_value(int v, int src_lane) { return __builtin_amdgcn_readlane(v, src_lane); }
__device__ __forceinline__ unsigned long long wave_sum64(unsigned long long v) {
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned lo = __shfl_xor((unsigned)v, off, kWave);
        const unsigned hi = __shfl_xor((unsigned)(v >> 32), off, kWave);
        v += ((unsigned long long)hi << 32) | lo;
    }
    return v;
}
__device__ __forceinline__ unsigned long long mix64(unsigned long long x) {  // splitmix64 finaliser
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// Phase timers of the profiling build (-DRAPID_PHASE_TIMERS, scripts/phase_timers.sh): shader-clock cycles per phase,
// summed over all waves, reported through the stats array instead of the usual counters.  Never in the product build.
#ifdef RAPID_PHASE_TIMERS
#define RAPID_T0(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#define RAPID_T1(acc, v) acc += __builtin_amdgcn_s_memtime() - (v)
#else
#define RAPID_T0(v)
#define RAPID_T1(acc, v)
#endif

// ---- detector state accessors ---------------------------------------------------------------------------------
// LDS flavour (population kernel): indices are slots; sweeps cover the hot slots only.
struct SlotDetector {
    unsigned int* st;  // one 32-bit word per hot slot (low 16 bits used): plain ds_or_rtn_b32, no sub-word shuffling
    const unsigned short* adj_off;
    const unsigned int* adj;
    int n_scan;  // = n_hot
    int H, L;
    unsigned int kmask;
    __device__ __forceinline__ unsigned int load(int i) const { return st[i]; }
    __device__ __forceinline__ void store(int i, unsigned int v) const { st[i] = v; }
    __device__ __forceinline__ int count(unsigned int m) const { return __popc(m & kmask); }
    __device__ __forceinline__ unsigned int or_bits(int i, unsigned int bits) const { return atomicOr(&st[i], bits); }
    __device__ __forceinline__ void clear_bits(int i, unsigned int bits) const { atomicAnd(&st[i], ~bits); }
    __device__ __forceinline__ void sync() const { wave_lds_fence(); }
};

// Global-memory flavour (one MultiNodeCutDetector instance, rapid_cd_*): indices are node indices, the implicit
// invalidation walks the view's observer table.  Accesses bypass the per-CU L1 (agent scope) because the atomics
// execute in L2; ordering points drain the vector memory queue.
struct TableDetector {
    unsigned short* st16;
    unsigned int* st32;
    const int* obs;  // [n_nodes][K]
    int n_scan;      // = n_nodes
    int K, H, L;
    unsigned int kmask;
    __device__ __forceinline__ unsigned int load(int i) const {
        return __hip_atomic_load(&st16[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __device__ __forceinline__ void store(int i, unsigned int v) const {
        __hip_atomic_store(&st16[i], (unsigned short)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __device__ __forceinline__ int count(unsigned int m) const { return __popc(m & kmask); }
    __device__ __forceinline__ unsigned int or_bits(int i, unsigned int bits) const {
        const int sh = (i & 1) * 16;
        return (atomicOr(&st32[i >> 1], bits << sh) >> sh) & 0xFFFFu;
    }
    __device__ __forceinline__ void sync() const { __syncthreads(); }
};

// Scalars of one receiver / detector (wave-uniform).
struct RxScalars {
    int running;         // updatesInProgress
    int npend;           // entries in pend[]
    int batch;           // batches fully processed
    int proposal_count;  // getNumProposals()
    bool seen_down;      // seenLinkDownEvents
    bool need_full;      // pend[] overflowed -> use the full pass over all hot slots
    bool batch_emitted;  // some emission happened in the batch being processed
};

// An emission (R/MultiNodeCutDetector.java:116-123): every entry that crossed H and was not yet returned is
// returned now and leaves `proposal`.  Marks them flushed; optionally appends them (ascending) to out[].
template <class D>
__device__ inline void flush_sweep(const D& d, int lane, int* out, int out_cap, int* out_n) {
    d.sync();
    for (int i0 = 0; i0 < d.n_scan; i0 += kWave) {
        const int i = i0 + lane;
        const unsigned int m = i < d.n_scan ? d.load(i) : 0u;
        const bool take = i < d.n_scan && d.count(m) >= d.H && !(m & kFlushed);
        if (take) d.store(i, m | kFlushed);
        if (out_n != nullptr) {
            const unsigned long long mk = wave_ballot(take);
            const int idx = *out_n + __popcll(mk & lanes_lt(lane));
            if (take && out != nullptr && idx < out_cap) out[idx] = i;
            *out_n += __popcll(mk);
        }
    }
    d.sync();
}

// EXACT application of one alert (all rings, ascending) -- R/MultiNodeCutDetector.java:76-128.  Executed
// redundantly by all lanes on wave-uniform values; lane 0 performs the stores.
template <class D>
__device__ inline void exact_apply(const D& d, RxScalars& s, unsigned short* pend, int dst, unsigned int bits, bool down,
                                   int lane, int* emit_out, int emit_cap, int* emit_n) {
    if (bits == 0) return;
    if (down) s.seen_down = true;
    unsigned int m = uniform(d.load(dst));
    unsigned int nb = bits & ~m & d.kmask;
    while (nb) {
        const int k = __ffs((int)nb) - 1;
        nb &= nb - 1;
        m |= 1u << k;
        const int c = d.count(m);
        if (c == d.L) {
            s.running++;
            if (pend != nullptr) {
                if (s.npend < kPendCap) {
                    if (lane == 0) pend[s.npend] = (unsigned short)dst;
                    s.npend++;
                } else {
                    s.need_full = true;
                }
            }
        }
        if (c == d.H) {
            s.running--;
            if (s.running == 0) {
                s.proposal_count++;
                s.batch_emitted = true;
                d.sync();
                if (lane == 0) d.store(dst, m);
                flush_sweep(d, lane, emit_out, emit_cap, emit_n);
                m = uniform(d.load(dst));
            }
        }
    }
    d.sync();
    if (lane == 0) d.store(dst, m);
    d.sync();
}

// Implicit-edge invalidation (R/MultiNodeCutDetector.java:137-164) along the round's hot adjacency.  An implicit
// report (o -> s, ring k) is applicable iff s is in preProposal and o is in proposal U preProposal; both require
// >= L explicit reports, so only hot slots take part, and a pair becomes applicable at the first batch end after
// BOTH ends crossed L -- i.e. when the later of the two is among the entrants since the previous pass.  One lane
// per entrant walks its adjacency list.  all_hot: the literal full pass (every hot slot), used after pend[]
// overflowed.  Returns the number of H crossings caused; logs every bit actually set when undo != nullptr.
__device__ inline int invalidate_adj(const SlotDetector& d, const unsigned short* pend, int n_ent, bool all_hot,
                                     unsigned int* undo, int* n_undo, int lane, int* n_applied) {
    int nH = 0;
    const int total = all_hot ? d.n_scan : n_ent;
    for (int i0 = 0; i0 < total; i0 += kWave) {
        const int i = i0 + lane;
        int e = 0;
        bool act = i < total;
        if (act) e = all_hot ? i : (int)pend[i];
        act = act && e < d.n_scan;
        int a = act ? (int)d.adj_off[e] : 0;
        const int end = act ? (int)d.adj_off[e + 1] : 0;
        unsigned int ent_next = a < end ? d.adj[a] : 0u;  // the (read-only) adjacency entry is fetched one step ahead
        while (wave_ballot(a < end) != 0ull) {
            const bool on = a < end;
            const unsigned int ent = ent_next;
            ent_next = a + 1 < end ? d.adj[a + 1] : 0u;
            const int other = (int)(ent & 0xFFFFu);
            const int k = (int)((ent >> 16) & 15u);
            const bool other_is_subject = ((ent >> 20) & 1u) != 0;
            const int sj = other_is_subject ? other : e;
            const int ob = other_is_subject ? e : other;
            const unsigned int ms = on ? d.load(sj) : 0u;
            const unsigned int mo = on ? d.load(ob) : 0u;
            const int cs = d.count(ms), co = d.count(mo);
            const bool apply = on && cs >= d.L && cs < d.H && co >= d.L && !(mo & kFlushed) && !(ms & (1u << k));
            unsigned int old = 0;
            if (apply) old = d.or_bits(sj, 1u << k);
            const bool isnew = apply && !(old & (1u << k));
            const bool crossH = isnew && d.count(old) == d.H - 1;
            nH += __popcll(wave_ballot(crossH));
            const unsigned long long mnew = wave_ballot(isnew);
            if (undo != nullptr) {
                const int idx = *n_undo + __popcll(mnew & lanes_lt(lane));
                if (isnew && idx < kUndoCap) undo[idx] = (unsigned)sj | ((unsigned)k << 24);
                *n_undo += __popcll(mnew);
            }
            *n_applied += __popcll(mnew);
            d.sync();
            ++a;
        }
    }
    return nH;
}

#if RAPID_FAST_WINDOW
// The same invalidation as ONE pass over the round's flat list of (subject slot, observer slot, ring) triples among the
// hot slots: pairs[0] = n, pairs[1 + a] = subject | observer << 14 | ring << 28.  Every triple is one potential implicit
// report; the test is invalidate_adj's.  A few hundred triples, 64 per step, whatever the number of entrants.
__device__ inline int invalidate_pairs(const SlotDetector& d, const unsigned int* pairs, unsigned int* undo, int* n_undo, int lane,
                                       int* n_applied) {
    int nH = 0;
    const int np = uniform((int)pairs[0]);
    for (int a0 = 0; a0 < np; a0 += kWave) {
        const int a = a0 + lane;
        const bool on = a < np;
        const unsigned int pr = on ? pairs[1 + a] : 0u;
        const int sj = (int)(pr & 0x3FFFu), ob = (int)((pr >> 14) & 0x3FFFu);
        const int k = (int)(pr >> 28);
        const unsigned int ms = on ? d.load(sj) : 0u, mo = on ? d.load(ob) : 0u;
        const int cs = d.count(ms), co = d.count(mo);
        const bool apply = on && cs >= d.L && cs < d.H && co >= d.L && !(mo & kFlushed) && !(ms & (1u << k));
        unsigned int old = 0;
        if (apply) old = d.or_bits(sj, 1u << k);
        const bool isnew = apply && !(old & (1u << k));
        const bool crossH = isnew && d.count(old) == d.H - 1;
        nH += __popcll(wave_ballot(crossH));
        const unsigned long long mnew = wave_ballot(isnew);
        if (undo != nullptr) {
            const int idx = *n_undo + __popcll(mnew & lanes_lt(lane));
            if (isnew && idx < kUndoCap) undo[idx] = (unsigned)sj | ((unsigned)k << 24);
            *n_undo += __popcll(mnew);
        }
        *n_applied += __popcll(mnew);
        d.sync();
    }
    return nH;
}
#endif

// The reference's literal pass over the view's observer table (single-detector API): every node in preProposal
// x its K observers (expected observers for a non-member).
__device__ inline int invalidate_table(const TableDetector& d, int lane) {
    int nH = 0;
    for (int n0 = 0; n0 < d.n_scan; n0 += kWave) {
        const int n = n0 + lane;
        const unsigned int m = n < d.n_scan ? d.load(n) : 0u;
        const int c = d.count(m);
        const bool inpre = n < d.n_scan && c >= d.L && c < d.H;
        if (wave_ballot(inpre) == 0ull) continue;
        for (int k = 0; k < d.K; ++k) {
            const int o = inpre ? d.obs[n * d.K + k] : -1;
            const unsigned int mo = o >= 0 ? d.load(o) : 0u;
            const bool apply = o >= 0 && d.count(mo) >= d.L && !(mo & kFlushed) && !(m & (1u << k));
            unsigned int old = 0;
            if (apply) old = d.or_bits(n, 1u << k);
            const bool isnew = apply && !(old & (1u << k));
            const bool crossH = isnew && d.count(old) == d.H - 1;
            nH += __popcll(wave_ballot(crossH));
        }
        d.sync();
    }
    return nH;
}

// EXACT end-of-batch step of the population kernel: invalidateFailingEdges as invoked at
// R/MembershipService.java:330.
__device__ inline void exact_batch_end(const SlotDetector& d, RxScalars& s, const unsigned short* pend, int lane,
                                       int* n_applied, int* n_full) {
    if (!s.seen_down) return;
    if (s.need_full) (*n_full)++;
    const int nH = invalidate_adj(d, pend, s.npend, s.need_full, nullptr, nullptr, lane, n_applied);
    s.npend = 0;
    s.running -= nH;
    if (nH > 0 && s.running == 0) {
        s.proposal_count++;
        s.batch_emitted = true;
        flush_sweep(d, lane, nullptr, 0, nullptr);
    }
}

// --------------------------------------------------------------------------------------------------------------
// Whole-population tally.  block = waves_per_block x 64; every wave pulls receivers until none is left.
// kTablesInLds: dictionary + adjacency are staged in LDS once per workgroup (the normal case); otherwise they
// are read from global memory (populations whose dictionary does not fit next to the per-wave state).
// --------------------------------------------------------------------------------------------------------------
// kTrusted: the engine has verified once, on the round's distinct alert set, that EVERY alert passes the filter of
// R/MembershipService.java:644-675 under the current view (configuration id, UP/DOWN vs membership, node range,
// non-empty ring list); the pipelined loop then skips the per-delivery re-check (the careful loop never does).
template <bool kTablesInLds, bool kTrusted>
__global__ __launch_bounds__(kMaxWavesPerBlock * 64) void tally_population_kernel(TallyParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = (int)(threadIdx.x & 63u);
    const int wave = (int)(threadIdx.x >> 6);

    // ---- shared read-only tables ----
    const unsigned short* dict = p.idx.dict;
    const unsigned short* adj_off = p.idx.adj_off;
    const unsigned int* adj = p.idx.adj;
    const int* node_of_slot = p.idx.node_of_slot;
    const unsigned short* subject_mask = nullptr;  // per hot slot, staged in LDS; computed on the fly otherwise
#if RAPID_FAST_WINDOW
    const unsigned int* pairs = nullptr;  // [0] = n, then (subject slot | observer slot << 14 | ring << 28); LDS tables only
#endif
    int shared_bytes = 0;
    if (kTablesInLds) {
        unsigned short* l_dict = reinterpret_cast<unsigned short*>(smem);
        unsigned short* l_off = reinterpret_cast<unsigned short*>(smem + align16(p.n_nodes * 2));
        unsigned int* l_adj =
            reinterpret_cast<unsigned int*>(smem + align16(p.n_nodes * 2) + align16((p.idx.n_hot + 1) * 2));
        for (int i = (int)threadIdx.x; i < p.n_nodes; i += (int)blockDim.x) l_dict[i] = p.idx.dict[i];
        for (int i = (int)threadIdx.x; i < p.idx.n_hot + 1; i += (int)blockDim.x) l_off[i] = p.idx.adj_off[i];
        for (int i = (int)threadIdx.x; i < p.idx.n_adj; i += (int)blockDim.x) l_adj[i] = p.idx.adj[i];
        int* l_nos = reinterpret_cast<int*>(smem + align16(p.n_nodes * 2) + align16((p.idx.n_hot + 1) * 2) + align16(p.idx.n_adj * 4));
        for (int i = (int)threadIdx.x; i < p.idx.n_hot; i += (int)blockDim.x) l_nos[i] = p.idx.node_of_slot[i];
        node_of_slot = l_nos;
        // rings on which a hot observer watches each hot slot: the implicit reports that slot can ever receive
        unsigned short* l_smask = reinterpret_cast<unsigned short*>(smem + align16(p.n_nodes * 2) + align16((p.idx.n_hot + 1) * 2) +
                                                                     align16(p.idx.n_adj * 4) + align16(p.idx.n_hot * 4));
        for (int i = (int)threadIdx.x; i < p.idx.n_hot; i += (int)blockDim.x) {
            unsigned int am = 0u;
            for (int a = (int)p.idx.adj_off[i]; a < (int)p.idx.adj_off[i + 1]; ++a) {
                const unsigned int ent = p.idx.adj[a];
                if (((ent >> 20) & 1u) == 0u) am |= 1u << ((ent >> 16) & 15u);  // role 0: slot i is the subject on that ring
            }
            l_smask[i] = (unsigned short)am;
        }
        subject_mask = l_smask;
#if RAPID_FAST_WINDOW
        // the hot adjacency as a flat list of (subject slot, observer slot, ring): one implicit report each
        unsigned int* l_pairs = reinterpret_cast<unsigned int*>(smem + align16(p.n_nodes * 2) + align16((p.idx.n_hot + 1) * 2) +
                                                                align16(p.idx.n_adj * 4) + align16(p.idx.n_hot * 4) + align16(p.idx.n_hot * 2));
        if (threadIdx.x == 0) l_pairs[0] = 0u;
        __syncthreads();
        for (int i = (int)threadIdx.x; i < p.idx.n_hot; i += (int)blockDim.x) {
            const int a0 = (int)p.idx.adj_off[i], a1 = (int)p.idx.adj_off[i + 1];
            int c = 0;
            for (int a = a0; a < a1; ++a) c += ((p.idx.adj[a] >> 20) & 1u) == 0u ? 1 : 0;
            if (c == 0) continue;
            unsigned int at = atomicAdd(&l_pairs[0], (unsigned int)c);
            for (int a = a0; a < a1; ++a) {
                const unsigned int ent = p.idx.adj[a];
                if (((ent >> 20) & 1u) == 0u) l_pairs[1 + at++] = (unsigned int)i | ((ent & 0x3FFFu) << 14) | (((ent >> 16) & 15u) << 28);
            }
        }
        pairs = l_pairs;
#endif
        dict = l_dict;
        adj_off = l_off;
        adj = l_adj;
        shared_bytes = tally_shared_bytes(p.n_nodes, p.idx.n_hot, p.idx.n_adj);
    }
    // The launch statistics are summed per workgroup in LDS and stored once per workgroup: thousands of waves adding to
    // the same eight global words at the end of their lives queue up behind each other in one L2 channel -- measured:
    // 0.13 ms of a 0.63 ms kernel, and every stream that crosses that channel waits with them.
    unsigned long long* const block_stats = reinterpret_cast<unsigned long long*>(
        smem + shared_bytes + (int)(blockDim.x >> 6) * tally_wave_bytes(p.idx.n_hot));
    if (threadIdx.x < 8u) block_stats[threadIdx.x] = 0ull;
    __syncthreads();

    // ---- this wave's private LDS ----
    const int state_bytes = align16(p.idx.n_hot * 4);
    unsigned char* const mine = smem + shared_bytes + wave * tally_wave_bytes(p.idx.n_hot);
    unsigned char* const stage = mine + state_bytes;
    unsigned short* const pend = reinterpret_cast<unsigned short*>(stage + kRingBytes);
    unsigned int* const undo = reinterpret_cast<unsigned int*>(stage + kRingBytes + align16(kPendCap * 2));
    const unsigned int* const ring32 = reinterpret_cast<const unsigned int*>(stage);
    const lds_addr_t ring_lds = lds_uniform(lds_address(stage));

    SlotDetector d;
    d.st = reinterpret_cast<unsigned int*>(mine);
    d.adj_off = adj_off;
    d.adj = adj;
    d.n_scan = p.idx.n_hot;
    d.H = p.H;
    d.L = p.L;
    d.kmask = (1u << p.K) - 1u;

    const unsigned int cfg_lo = (unsigned int)(unsigned long long)p.cfg_id;
    const unsigned int cfg_hi = (unsigned int)((unsigned long long)p.cfg_id >> 32);
    unsigned long long n_slow = 0, n_fast = 0, n_restart = 0, n_records = 0, n_pipe = 0, n_careful = 0;
#ifdef RAPID_PHASE_TIMERS
    unsigned long long t_total = 0, t_ensure = 0, t_lean = 0, t_careful = 0, t_out = 0, t_flush = 0, t_rx = 0;
    RAPID_T0(t_kernel0);
#endif
    int n_applied = 0, n_full = 0;
    const int lane20 = lane * kRecBytes;

    const unsigned int lane16 = (unsigned int)lane * 16u;
    // the stream of a receiver as the ring sees it
    struct Stream {
        dma_rsrc_t rsrc;
        int delta, nrec;
    };
    // The ring starts 0..3 records BEFORE the receiver's first one, at the 16-B aligned address a0 = b0 - 20 m with
    // m = rec0 mod 4: stream byte 20 (i + m) holds record i, and since the ring is a whole number of records (512) no
    // record ever straddles its end.
    auto make_stream = [&](long long rec0, long long rec1) -> Stream {
        Stream st;
        st.nrec = (int)(rec1 - rec0);
        st.delta = (int)(rec0 & 3) * kRecBytes;
        const unsigned long long a0 = (unsigned long long)rec0 * kRecBytes - (unsigned long long)st.delta;
        unsigned long long span = ((unsigned long long)st.delta + (unsigned long long)st.nrec * kRecBytes + 15ull) & ~15ull;
        if (a0 + span > p.records_bytes) span = p.records_bytes > a0 ? p.records_bytes - a0 : 0ull;
        st.rsrc = dma_make_rsrc(p.records + a0, (unsigned int)(span > 0xFFFFFFF0ull ? 0xFFFFFFF0ull : span));
        return st;
    };
    auto uniform64 = [&](long long v) -> long long {
        return ((long long)uniform((unsigned int)((unsigned long long)v >> 32)) << 32) | (long long)uniform((unsigned int)(unsigned long long)v);
    };
    auto issue_head = [&](const Stream& st) {  // the first kDepth KiB of a stream into ring slots 0 .. kDepth - 1
        const dma_rsrc_t head_rsrc = dma_uniform(st.rsrc);
#pragma unroll
        for (int k = 0; k < kDepth; ++k) lds_dma16(head_rsrc, lane16, (unsigned int)k * kSlotBytes, ring_lds + k * kSlotBytes);
    };

    // Receivers are dealt round-robin: wave g of G takes receivers g, g + G, g + 2 G ...  Every receiver of a round costs
    // about the same (they all see the same alerts), so a shared work counter would balance nothing -- but 2,560 waves
    // hitting one counter at the same moment queue up behind each other (measured: ~15 us per receiver waiting for the
    // atomic), and a static deal lets the next stream's bounds be loaded a whole receiver ahead.  Everything the
    // compiler has to wait for is waited for at ONE point per receiver -- right after the stream of r has drained,
    // when none of the asm-issued loads is in flight -- because its wait for any of its own loads is a full
    // s_waitcnt vmcnt(0).
    // wave-major numbering: consecutive receivers go to different CUs, so the last, partial round still uses every CU
    const int wave_global = uniform(wave * (int)gridDim.x + (int)blockIdx.x), waves_total = (int)gridDim.x * (int)(blockDim.x >> 6);
#if RAPID_EARLY_CERT
    // with no report at all, can any hot slot reach H through implicit reports alone?  (constant for the round)
    bool pot_below_h = kTablesInLds;
    if (kTablesInLds) {
        unsigned long long any_high0 = 0ull;
        for (int i0 = 0; i0 < p.idx.n_hot; i0 += kWave) {
            const int i = i0 + lane;
            any_high0 |= wave_ballot(i < p.idx.n_hot && __popc((unsigned int)subject_mask[i < p.idx.n_hot ? i : 0] & ((1u << p.K) - 1u)) >= p.H);
        }
        pot_below_h = any_high0 == 0ull;
    }
#endif
    int r = wave_global;
#ifdef RAPID_REVERSE_DEAL  // measurement aid: the same deal over the receivers in reverse order
#define RAPID_RX(i) (p.n_receivers - 1 - (i))
#else
#define RAPID_RX(i) (i)
#endif
    Stream cur = make_stream(0, 0);
    if (r < p.n_receivers) cur = make_stream(p.rec_off[RAPID_RX(r)], p.rec_off[RAPID_RX(r) + 1]);
    bool prestarted = false;  // the head of `cur` is already on its way into the ring
    while (r < p.n_receivers) {
#ifdef RAPID_PHASE_TIMERS
        const unsigned long long t_rx0 = __builtin_amdgcn_s_memtime();
        const unsigned long long t_ensure0 = t_ensure, t_lean0 = t_lean, t_careful0 = t_careful;
#ifdef RAPID_TIMER_REALTIME
        const unsigned long long t_real0 = __builtin_amdgcn_s_memrealtime();  // constant 100 MHz
#endif
#endif
        const int r_next = uniform(r + waves_total);
        long long n0_v = 0, n1_v = 0;  // lane 0: stream bounds of r_next, consumed after this receiver's stream
        if (lane == 0 && r_next < p.n_receivers) {
            n0_v = p.rec_off[RAPID_RX(r_next)];
            n1_v = p.rec_off[RAPID_RX(r_next) + 1];
        }
        const dma_rsrc_t rsrc = dma_uniform(cur.rsrc);
        const int delta = uniform(cur.delta), nrec = uniform(cur.nrec);

        int emit_batch = -1;
        RxScalars s;
        bool exact_only = (p.flags & 1) != 0;
        bool restart = true;  // (re)initialise the detector before the first sub-chunk
        int pos = 0;          // next unconsumed record
        int ring_pos = delta;  // its byte offset in the ring: (delta + 20 pos) mod kRingBytes
        auto advance = [&](int n) {  // n <= kLeanWindow records consumed (less than the ring: one wrap at most)
            pos += n;
            ring_pos += n * kRecBytes;
            if (ring_pos >= kRingBytes) ring_pos -= kRingBytes;
        };
        int careful_budget = 0, careful_next = 1;
        int careful_cap = kWave;  // records the careful loop takes at once; halved while an emission cannot be excluded
#if RAPID_CAREFUL_HINT
        int hint_cap = -1;        // after a failed attempt: records of the sub-chunk before its critical H crossing (-1: unknown)
        bool exact_next = false;  // the next sub-chunk starts with the critical record: replay it record by record
#endif

        // per-sub-chunk decode results (one record per lane)
        int dst = 0, ncons = 0, lastE = -1;
        unsigned int bits = 0;
        bool down = false, eob = false, hasadj = false;
        unsigned long long mE_all = 0ull;

        // dword `i` of the record that starts at ring byte offset t (t < 2 * kRingBytes, already wrapped if record-aligned)
        auto ring_word = [&](unsigned int t, int i) -> unsigned int {
            if (kRingRecordAligned) return ring32[(t >> 2) + i];
            unsigned int x = t + 4u * (unsigned int)i;
            x = min(x, x - (unsigned int)kRingBytes);
            return ring32[x >> 2];
        };
        // ---- decode the sub-chunk starting at `pos` from the LDS ring ----
        auto decode = [&]() {
            const int navail = min(careful_cap, nrec - pos);
            unsigned int t = (unsigned int)(ring_pos + lane20);
            if (kRingRecordAligned) t = min(t, t - (unsigned int)kRingBytes);  // per-lane wrap
            const unsigned int w0 = ring_word(t, 0), w1 = ring_word(t, 1), w3 = ring_word(t, 3), w4 = ring_word(t, 4);
            down = ((w4 >> 16) & 0xFFu) != 0;
            eob = lane < navail && ((((w4 >> 24) & 1u) != 0) || pos + lane == nrec - 1);
            // a sub-chunk ends at its last batch end (if it has one): no record is applied before the batch end
            // that precedes it has been processed
            mE_all = wave_ballot(eob);
            lastE = mE_all ? 63 - __clzll((long long)mE_all) : -1;
            ncons = lastE >= 0 ? lastE + 1 : navail;
            const bool valid = lane < ncons;
            eob = eob && valid;
            // filterAlertMessages (R/MembershipService.java:644-675) + node -> slot, branch-free
            const unsigned int de = (unsigned int)dict[w3 < (unsigned)p.n_nodes ? w3 : 0u];
            const bool okf = valid & (w0 == cfg_lo) & (w1 == cfg_hi) & (w3 < (unsigned)p.n_nodes) &
                             (((de & kDictMember) != 0) == down) & ((w4 & d.kmask) != 0u);
            const bool hot = (de & kSlotMask) != kNoSlot;
            down = down & okf;  // from here on: "a DOWN report passed the filter" (sets seenLinkDownEvents)
            dst = (okf & hot) ? (int)(de & kSlotMask) : 0;
            hasadj = (de & kDictHasAdj) != 0;
            bits = (okf & hot) ? (w4 & d.kmask) : 0u;
        };

        // ---- apply the deferred implicit invalidation for the entrants queued by the lean path.  Every report applied
        // here was applied by the reference at a batch end inside a window already certified emission-free, and the
        // certificate's witness (see below) takes no implicit reports, so this cannot be where an emission happens.
#if RAPID_FAST_WINDOW
        bool owed_reports = false;  // fast windows were applied since the last flush: the implicit reports they make possible are owed
        bool repicked = false;    // the early witness has been exchanged for the best one available
#endif
        auto flush_pending = [&]() {
#if RAPID_FAST_WINDOW
            if (!s.seen_down || s.need_full || (s.npend == 0 && !owed_reports)) return;
#ifdef RAPID_TRACE
            if (lane == 0) fprintf(stderr, "F r=%d pos=%d npend=%d owed=%d\n", r, pos, s.npend, (int)owed_reports);
#endif
            wave_lds_fence();
            int applied = 0;
            if (kTablesInLds) {
                (void)invalidate_pairs(d, pairs, nullptr, nullptr, lane, &applied);
            } else {
                (void)invalidate_adj(d, pend, 0, true, nullptr, nullptr, lane, &applied);  // the same pass over the adjacency lists
            }
            s.npend = 0;
            owed_reports = false;
            n_applied += applied;
#else
            if (s.npend == 0 || !s.seen_down || s.need_full) return;
#ifdef RAPID_TRACE
            if (lane == 0) fprintf(stderr, "F r=%d pos=%d npend=%d\n", r, pos, s.npend);
#endif
            wave_lds_fence();
            int applied = 0;
            (void)invalidate_adj(d, pend, s.npend, false, nullptr, nullptr, lane, &applied);
            s.npend = 0;
            n_applied += applied;
#endif
        };

        // ---- one pass over the hot slots: updatesInProgress (slots with L <= count < H) as the careful path needs it,
        // and a WITNESS for the lean path: a slot in [L, H) that stays below H even if it is given every implicit report
        // it can ever get -- popc(state | rings on which a hot observer watches it) < H -- so that a lagging state cannot
        // hide its departure; the one with the smallest such bound.  -1 if there is none.
        int witness = -1;
        unsigned int witness_mask = 0u;  // rings on which the witness can receive an implicit report
#if RAPID_EARLY_CERT
        bool below_h = false;  // every slot has popc(state | subject_mask) < H: known exactly, nothing applied unchecked since
#endif
        bool running_exact = false;      // s.running is the reference's updatesInProgress (nothing queued, kept up to date)
#if RAPID_CAREFUL_HINT
        int n_at_h = 0;  // slots that have reached H, as of the last recount
#endif
        auto recount = [&]() {
#ifdef RAPID_TRACE
            if (lane == 0) fprintf(stderr, "R r=%d pos=%d\n", r, pos);
#endif
            wave_lds_fence();
            int run = 0;
#if RAPID_EARLY_CERT
            unsigned long long any_high = 0ull;
#endif
#if RAPID_CAREFUL_HINT
            n_at_h = 0;
#endif
            unsigned int best = 0xFFFFFFFFu, best_mask = 0u;  // bound << 16 | slot
            for (int i0 = 0; i0 < d.n_scan; i0 += kWave) {
                const int i = i0 + lane;
                const unsigned int m = i < d.n_scan ? d.load(i) : 0u;
                const int c = d.count(m);
                const bool pre = i < d.n_scan && c >= d.L && c < d.H;
                run += __popcll(wave_ballot(pre));
#if RAPID_EARLY_CERT
                if (kTablesInLds) any_high |= wave_ballot(i < d.n_scan && d.count(m | (unsigned int)subject_mask[i < d.n_scan ? i : 0]) >= d.H);
#endif
#if RAPID_CAREFUL_HINT
                n_at_h += __popcll(wave_ballot(i < d.n_scan && c >= d.H));
#endif
                unsigned int am = 0u;
                if (kTablesInLds) {
                    am = pre ? (unsigned int)subject_mask[i] : 0u;
                } else {
                    int a = pre ? (int)adj_off[i] : 0;
                    const int a_end = pre ? (int)adj_off[i + 1] : 0;
                    while (wave_ballot(a < a_end) != 0ull) {
                        const unsigned int ent = a < a_end ? adj[a] : (1u << 20);
                        if (((ent >> 20) & 1u) == 0u) am |= 1u << ((ent >> 16) & 15u);  // role 0: slot i is the subject on that ring
                        ++a;
                    }
                }
                const int bound = d.count(m | am);
                const unsigned int key = (pre && bound < d.H) ? ((unsigned int)bound << 16) | (unsigned int)i : 0xFFFFFFFFu;
                if (key < best) {
                    best = key;
                    best_mask = am;
                }
            }
            unsigned int all_best = best;
            for (int o2 = 32; o2 > 0; o2 >>= 1) all_best = min(all_best, (unsigned int)__shfl_xor((int)all_best, o2, kWave));
            all_best = uniform(all_best);
            s.running = run;
            running_exact = s.npend == 0;  // with nothing queued the state here is the reference's
#if RAPID_FAST_WINDOW
            running_exact = running_exact && !owed_reports;
            repicked = pos >= kRepickAfter;  // this WAS the sweep for the best witness, unless it came too early to count
#endif
#if RAPID_EARLY_CERT
            below_h = kTablesInLds && any_high == 0ull;
#endif
            if (all_best == 0xFFFFFFFFu) {
                witness = -1;
                witness_mask = 0u;
            } else {
                witness = (int)(all_best & 0xFFFFu);
                const unsigned long long holder = wave_ballot(best == all_best);
                witness_mask = (unsigned int)lane_value((int)best_mask, __ffsll((long long)holder) - 1);
            }
        };

        // ---- LEAN path: order-free application of a window of up to kLeanWindow records, kQuarters per lane (part q =
        // records pos + 64 q + lane), cut at its last batch end; the implicit invalidation is deferred (entrants are queued in
        // pend[]).  The window is committed only with a CERTIFICATE that the reference cannot emit at any point inside
        // it.  An emission needs updatesInProgress to reach 0 (R/MultiNodeCutDetector.java:110-121), so either of these
        // suffices:
        //   (a) the witness -- in preProposal before the window -- stays below H even when credited with every implicit
        //       report it can ever get, popc(state | witness_mask) < H after the window: it was in preProposal
        //       throughout, updatesInProgress >= 1 (the bound covers whatever the deferred invalidation still owes it);
        //   (b) no report of the window takes any subject to H, and no implicit report can be generated in it (nothing
        //       queued, no entrant with hot adjacency): then nothing crosses H at all;
        //   (c) as (b), but with fewer H crossings than an exactly known updatesInProgress.
        // Counts only grow, LDS atomics of one wave execute in program order (a later quarter sees the earlier ones'
        // bits), and every crossing of L is seen by exactly one lane, whatever the order.  The parts are independent
        // dependency chains for the wave to interleave.  Returns 0 -- window rolled back,
        // nothing consumed -- when no certificate holds.  kTail: fewer than kLeanWindow records are left (the last record
        // closes the last batch).
        auto lean_window = [&](auto tail_tag, auto seen_tag) -> int {
            constexpr bool kTail = decltype(tail_tag)::value;
            constexpr bool kSeen = decltype(seen_tag)::value;  // the caller knows that s.seen_down is set already
            const int navail = kTail ? nrec - pos : kLeanWindow;
            unsigned int w3[kQuarters], w4[kQuarters], rb[kQuarters], de[kQuarters], slot[kQuarters], old[kQuarters];
            unsigned long long mE[kQuarters], mApp[kQuarters], mL[kQuarters], mJ[kQuarters];
            int nc[kQuarters];
#pragma unroll
            for (int q = 0; q < kQuarters; ++q) {
                unsigned int t = (unsigned int)(ring_pos + lane20 + q * kWave * kRecBytes);
                if (kRingRecordAligned) t = min(t, t - (unsigned int)kRingBytes);  // per-lane wrap
                w3[q] = ring_word(t, 3);
                w4[q] = ring_word(t, 4);
            }
            unsigned long long anyE = 0ull;
#pragma unroll
            for (int q = 0; q < kQuarters; ++q) {
                if (kTail)
                    mE[q] = wave_ballot((lane + q * kWave < navail) & (((w4[q] & 0x01000000u) != 0u) | (lane + q * kWave == navail - 1)));
                else
                    mE[q] = wave_ballot((w4[q] & 0x01000000u) != 0u);
                anyE |= mE[q];
            }
            // consume up to the last batch end of the window (none at all: a batch longer than the window -> careful path)
            int consumed = 0;
            {
                bool later = false;  // a later quarter has a batch end: this one is consumed whole
#pragma unroll
                for (int q = kQuarters - 1; q >= 0; --q) {
                    nc[q] = later ? kWave : (mE[q] != 0ull ? kWave - __clzll((long long)mE[q]) : 0);
                    later = later || mE[q] != 0ull;
                    consumed += nc[q];
                }
            }
            const bool seen_before = kSeen || s.seen_down;
            unsigned long long mDown = 0ull;
#if RAPID_LEAN_V2
            bool appl[kQuarters];
#endif
#pragma unroll
            for (int q = 0; q < kQuarters; ++q) {
                rb[q] = w4[q] & d.kmask;
                bool app;
                if (kTrusted) {
                    // every consumed record is a validated alert: the dictionary lookup needs no range check
                    de[q] = (unsigned int)dict[(!kTail || lane < nc[q]) ? w3[q] : 0u];
                    app = (lane < nc[q]) & ((de[q] & kSlotMask) != kNoSlot);
#if RAPID_LEAN_V2
                    mApp[q] = wave_ballot(lane < nc[q]) & wave_ballot((de[q] & kSlotMask) != kNoSlot);
                    if (!seen_before) mDown |= wave_ballot(lane < nc[q]) & wave_ballot((w4[q] & 0x00FF0000u) != 0u);
#else
                    if (!seen_before) mDown |= wave_ballot((lane < nc[q]) & ((w4[q] & 0x00FF0000u) != 0u));
#endif
                } else {
                    // filterAlertMessages (R/MembershipService.java:644-675) + node -> slot
                    unsigned int t = (unsigned int)(ring_pos + lane20 + q * kWave * kRecBytes);
                    if (kRingRecordAligned) t = min(t, t - (unsigned int)kRingBytes);
                    const unsigned int w0 = ring_word(t, 0), w1 = ring_word(t, 1);
                    de[q] = (unsigned int)dict[w3[q] < (unsigned)p.n_nodes ? w3[q] : 0u];
                    const unsigned int dn = (w4[q] & 0x00FF0000u) != 0u ? 1u : 0u;
                    const unsigned int bad = (w0 ^ cfg_lo) | (w1 ^ cfg_hi) | (dn ^ (de[q] >> 15)) |
                                             (w3[q] >= (unsigned)p.n_nodes ? 1u : 0u) | (rb[q] == 0u ? 1u : 0u) |
                                             (lane >= nc[q] ? 1u : 0u);
                    app = (bad == 0u) & ((de[q] & kSlotMask) != kNoSlot);
#if RAPID_LEAN_V2
                    mApp[q] = wave_ballot(bad == 0u) & wave_ballot((de[q] & kSlotMask) != kNoSlot);
#endif
                    if (!seen_before) mDown |= wave_ballot((bad | (dn ^ 1u)) == 0u);
                }
                slot[q] = de[q] & kSlotMask;
#if RAPID_LEAN_V2
                appl[q] = app;
#else
                mApp[q] = wave_ballot(app);  // all lane masks first: the atomics then go out back to back
#endif
            }
            if (!seen_before) s.seen_down = mDown != 0ull;
#pragma unroll
            for (int q = 0; q < kQuarters; ++q) {
                old[q] = 0u;
#if RAPID_LEAN_V2
                if (appl[q]) old[q] = d.or_bits((int)slot[q], rb[q]);
#else
                if ((mApp[q] >> lane) & 1ull) old[q] = d.or_bits((int)slot[q], rb[q]);
#endif
            }
            // The witness's reports AFTER the window: a wave's LDS operations are served in program order, so this read
            // sees every lane's atomic above without waiting for their results (the barrier is not an instruction; it
            // keeps the compiler -- and the lane-by-lane emulator -- from moving the read ahead of them).
            __builtin_amdgcn_wave_barrier();
            const unsigned int wv = uniform(d.load(witness >= 0 ? witness : 0));
            unsigned long long anyX = 0ull, anyN = 0ull;
            int nX = 0;
#if RAPID_LEAN_V2
            bool isX[kQuarters];  // this lane's record is an entrant whose implicit reports are owed
#endif
#pragma unroll
            for (int q = 0; q < kQuarters; ++q) {
                const unsigned int ok = old[q] & d.kmask;
                const int c0 = __popc(ok), c1 = __popc(ok | rb[q]);
                mL[q] = mApp[q] & wave_ballot(c0 < d.L) & wave_ballot(c1 >= d.L);
#if RAPID_LEAN_V2
                isX[q] = appl[q] & (c0 < d.L) & (c1 >= d.L) & ((de[q] & kDictHasAdj) != 0u);
#endif
                // entrants with hot adjacency (their implicit reports are owed) / without (witness material)
                mJ[q] = wave_ballot((de[q] & kDictHasAdj) != 0u);
                anyX |= mL[q] & mJ[q];
                anyN |= mL[q] & ~mJ[q];
                nX += __popcll(mL[q] & mJ[q]);
            }
            bool certified = witness >= 0 && __popc((wv | witness_mask) & d.kmask) < d.H;
            int nLc = 0, nHc = 0;
            bool counted = false;
#if RAPID_EARLY_CERT
            bool by_bound = false;
            unsigned int smq[kQuarters];
            if (kTablesInLds && witness < 0 && below_h) {
                unsigned long long mHigh = 0ull;
#pragma unroll
                for (int q = 0; q < kQuarters; ++q) {
                    const bool mine_ = ((mApp[q] >> lane) & 1ull) != 0ull;
                    smq[q] = mine_ ? (unsigned int)subject_mask[slot[q]] : 0u;
                    mHigh |= mApp[q] & wave_ballot(__popc((old[q] | rb[q] | smq[q]) & d.kmask) >= d.H);
                }
                by_bound = mHigh == 0ull;
                certified = by_bound;
                below_h = by_bound;
            }
#endif
            if (!certified) {
                // (b)/(c): no implicit report can be generated inside the window (nothing queued, no entrant with hot
                // adjacency), so only its explicit reports cross H -- none of them, or fewer than updatesInProgress
                unsigned long long mH = 0ull;
#pragma unroll
                for (int q = 0; q < kQuarters; ++q) {
                    const unsigned int ok = old[q] & d.kmask;
                    const unsigned long long mHq = mApp[q] & wave_ballot(__popc(ok) < d.H) & wave_ballot(__popc(ok | rb[q]) >= d.H);
                    mH |= mHq;
                    nHc += __popcll(mHq);
                    nLc += __popcll(mL[q]);
                }
                counted = true;
                certified = s.npend == 0 && nX == 0 && (mH == 0ull || (running_exact && s.running - nHc >= 1));
#if RAPID_FAST_WINDOW
                certified = certified && !owed_reports;  // entrants of fast windows are owed their implicit reports too
#endif
            }
            if (__builtin_expect(!certified || anyE == 0ull || s.npend + nX > kPendCap, 0)) {
#ifdef RAPID_TRACE
                if (lane == 0) fprintf(stderr, "L-fail r=%d pos=%d witness=%d wcount=%d npend=%d nX=%d noE=%d\n", r, pos, witness, __popc(wv & d.kmask), s.npend, nX, (int)(anyE == 0ull));
#endif
#pragma unroll
                for (int q = 0; q < kQuarters; ++q) {
                    const unsigned int fresh = rb[q] & ~old[q];
                    if (((mApp[q] >> lane) & 1ull) && fresh != 0u) d.clear_bits((int)slot[q], fresh);
                }
                s.seen_down = seen_before;
                wave_lds_fence();
                return 0;
            }
#ifdef RAPID_TRACE
            if (lane == 0) fprintf(stderr, "L-ok r=%d pos=%d nc=%d witness=%d wcount=%d npend=%d nX=%d batch=%d\n", r, pos, consumed, witness, __popc(wv & d.kmask), s.npend, nX, s.batch);
#endif
#if RAPID_EARLY_CERT
            if (by_bound) {
                // an entrant of this window is in preProposal from here on and cannot reach H whatever it is credited with
#pragma unroll
                for (int q = 0; q < kQuarters; ++q)
                    if (mL[q] != 0ull) {
                        const int src = __ffsll((long long)mL[q]) - 1;
                        witness = lane_value((int)slot[q], src);
                        witness_mask = (unsigned int)lane_value((int)smq[q], src);
                    }
            } else {
                below_h = false;  // applied without looking at the bounds
            }
#endif
            if (anyX != 0ull) {  // queue the entrants whose implicit reports are owed
                int base = s.npend;
#pragma unroll
                for (int q = 0; q < kQuarters; ++q) {
                    const unsigned long long mX = mL[q] & mJ[q];
#if RAPID_LEAN_V2
                    if (isX[q]) pend[base + rank_below(mX)] = (unsigned short)slot[q];
#else
                    if ((mX >> lane) & 1ull) pend[base + __popcll(mX & lanes_lt(lane))] = (unsigned short)slot[q];
#endif
                    base += __popcll(mX);
                }
                s.npend = base;
            }
            // a fresh entrant without hot adjacency is the best witness there is: it needs H - L more reports to leave
            if (anyN != 0ull) {
#pragma unroll
                for (int q = 0; q < kQuarters; ++q) {
                    const unsigned long long mN = mL[q] & ~mJ[q];
                    if (mN != 0ull) witness = lane_value((int)slot[q], __ffsll((long long)mN) - 1);
                }
                witness_mask = 0u;
            }
            // updatesInProgress stays exact only through windows that counted their crossings
            if (counted && running_exact)
                s.running += nLc - nHc;
            else
                running_exact = false;
            int nbatches = 0;
#pragma unroll
            for (int q = 0; q < kQuarters; ++q) nbatches += __popcll(mE[q]);
            s.batch += nbatches;
            advance(consumed);
            return 1;
        };

#if RAPID_FAST_WINDOW
        // ---- FAST window: kLeanWindow records, a witness exists, a DOWN report has been seen.  Certificate (a) of the lean
        // window, evaluated BEFORE anything is applied: the witness's state after the window is its state now plus the
        // bits of the window's records about it.  Returns 0 with nothing applied when the witness could reach H (or the
        // window holds no batch end).
        auto fast_window = [&]() -> int {
            unsigned int w3[kQuarters], w4[kQuarters], rb[kQuarters], slot[kQuarters];
            bool app[kQuarters];
            unsigned long long mE[kQuarters], mApp[kQuarters];
            int nc[kQuarters];
#pragma unroll
            for (int q = 0; q < kQuarters; ++q) {
                unsigned int t = (unsigned int)(ring_pos + lane20 + q * kWave * kRecBytes);
                if (kRingRecordAligned) t = min(t, t - (unsigned int)kRingBytes);  // per-lane wrap
                w3[q] = ring_word(t, 3);
                w4[q] = ring_word(t, 4);
            }
            unsigned long long anyE = 0ull;
#pragma unroll
            for (int q = 0; q < kQuarters; ++q) {
                mE[q] = wave_ballot((w4[q] & 0x01000000u) != 0u);
                anyE |= mE[q];
            }
            if (anyE == 0ull) return 0;  // a batch longer than the window: the careful path takes it
            int consumed = 0;
            {
                bool later = false;
#pragma unroll
                for (int q = kQuarters - 1; q >= 0; --q) {
                    nc[q] = later ? kWave : (mE[q] != 0ull ? kWave - __clzll((long long)mE[q]) : 0);
                    later = later || mE[q] != 0ull;
                    consumed += nc[q];
                }
            }
            unsigned int wadd = 0u;  // what the window reports about the witness
#pragma unroll
            for (int q = 0; q < kQuarters; ++q) {
                rb[q] = w4[q] & d.kmask;
                unsigned int de;
                if (kTrusted) {
                    de = (unsigned int)dict[w3[q]];  // every record of a full window is a validated alert
                    app[q] = (lane < nc[q]) & ((de & kSlotMask) != kNoSlot);
                    mApp[q] = wave_ballot(lane < nc[q]) & wave_ballot((de & kSlotMask) != kNoSlot);
                } else {
                    unsigned int t = (unsigned int)(ring_pos + lane20 + q * kWave * kRecBytes);
                    if (kRingRecordAligned) t = min(t, t - (unsigned int)kRingBytes);
                    const unsigned int w0 = ring_word(t, 0), w1 = ring_word(t, 1);
                    de = (unsigned int)dict[w3[q] < (unsigned)p.n_nodes ? w3[q] : 0u];
                    const unsigned int dn = (w4[q] & 0x00FF0000u) != 0u ? 1u : 0u;
                    const unsigned int bad = (w0 ^ cfg_lo) | (w1 ^ cfg_hi) | (dn ^ (de >> 15)) |
                                             (w3[q] >= (unsigned)p.n_nodes ? 1u : 0u) | (rb[q] == 0u ? 1u : 0u) |
                                             (lane >= nc[q] ? 1u : 0u);
                    app[q] = (bad == 0u) & ((de & kSlotMask) != kNoSlot);
                    mApp[q] = wave_ballot(bad == 0u) & wave_ballot((de & kSlotMask) != kNoSlot);
                }
                slot[q] = de & kSlotMask;
            }
            const unsigned int wv = uniform(d.load(witness));
            {
                unsigned long long mW[kQuarters], anyW = 0ull;
#pragma unroll
                for (int q = 0; q < kQuarters; ++q) {
                    mW[q] = mApp[q] & wave_ballot(slot[q] == (unsigned int)witness);
                    anyW |= mW[q];
                }
                if (anyW != 0ull) {  // rare: the witness receives ten reports in a whole stream
#pragma unroll
                    for (int q = 0; q < kQuarters; ++q)
                        for (unsigned long long m = mW[q]; m != 0ull; m &= m - 1ull)
                            wadd |= (unsigned int)lane_value((int)rb[q], __ffsll((long long)m) - 1);
                }
            }
            if (__popc((wv | wadd | witness_mask) & d.kmask) >= d.H) {
#ifdef RAPID_TRACE
                if (lane == 0) fprintf(stderr, "W-fail r=%d pos=%d witness=%d wcount=%d\n", r, pos, witness, __popc((wv | wadd) & d.kmask));
#endif
                return 0;
            }
#pragma unroll
            for (int q = 0; q < kQuarters; ++q)
                if (__builtin_expect(app[q], 1)) (void)d.or_bits((int)slot[q], rb[q]);
            owed_reports = true;
            running_exact = false;
#if RAPID_EARLY_CERT
            below_h = false;
#endif
            int nbatches = 0;
#pragma unroll
            for (int q = 0; q < kQuarters; ++q) nbatches += __popcll(mE[q]);
            s.batch += nbatches;
            advance(consumed);
            return 1;
        };
#endif

        // ---- CAREFUL path (non-pipelined loop), first attempt: order-free application with the implicit
        // invalidation applied immediately and an EXACT count of the H crossings; rolled back if an emission
        // cannot be excluded.  Returns false when the sub-chunk must be replayed record by record. ----
        auto immediate_subchunk = [&]() -> bool {
            unsigned int old = 0;
            if (bits) old = d.or_bits(dst, bits);
            const unsigned int newbits = bits & ~old;
            const int c0 = d.count(old), c1 = d.count(old | bits);
            const bool isL = bits != 0 && c0 < d.L && c1 >= d.L;
            const bool isH = bits != 0 && c0 < d.H && c1 >= d.H;
            const unsigned long long mL = wave_ballot(isL), mH = wave_ballot(isH);
            const unsigned long long mD = wave_ballot(down);
            const int nLc = __popcll(mL), nHc = __popcll(mH);
            const bool seen = s.seen_down || mD != 0ull;
            bool need_full = s.need_full;
            const int posn = s.npend + __popcll(mL & lanes_lt(lane));
            if (isL && posn < kPendCap) pend[posn] = (unsigned short)dst;
            const int npend_new = s.npend + nLc;
            if (npend_new > kPendCap) need_full = true;
            const bool run_inv = lastE >= 0 && seen && (npend_new > 0 || need_full);
            int nHi = 0, n_undo = 0, applied_here = 0;
            if (run_inv) {
                wave_lds_fence();
                if (need_full) n_full++;
#if RAPID_FAST_WINDOW
                if (kTablesInLds)
                    nHi = invalidate_pairs(d, pairs, undo, &n_undo, lane, &applied_here);
                else
#endif
                    nHi = invalidate_adj(d, pend, npend_new, need_full, undo, &n_undo, lane, &applied_here);
            }
            const int Htot = nHc + nHi;
            if (Htot == 0 || s.running - Htot >= 1) {
                s.running += nLc - Htot;
                s.seen_down = seen;
                s.batch += __popcll(mE_all);
                s.need_full = need_full;
                s.npend = ((lastE >= 0 && seen) || need_full) ? 0 : npend_new;
                n_applied += applied_here;
                n_fast++;
                n_records += (unsigned long long)ncons;
                advance(ncons);
                return true;
            }
#if RAPID_CAREFUL_HINT
            // which record is critical: the one with the running-th explicit H crossing (the records before it hold fewer
            // crossings than updatesInProgress).  Only a hint for the size of the next attempt.
            hint_cap = -1;
            if (nHi == 0 && s.running >= 1 && s.running <= 8 && nHc >= s.running) {
                unsigned long long m = mH;
                for (int i = 1; i < s.running; ++i) m &= m - 1ull;
                hint_cap = __ffsll((long long)m) - 1;
            }
#endif
            if (n_undo > kUndoCap) {
                restart = true;  // cannot roll back: redo this receiver on the exact path only
                exact_only = true;
                n_restart++;
                return false;
            }
            wave_lds_fence();
            for (int u = lane; u < n_undo; u += kWave) {
                const unsigned int e = undo[u];
                d.clear_bits((int)(e & 0xFFFFFFu), 1u << (e >> 24));
            }
            if (newbits) d.clear_bits(dst, newbits);
            wave_lds_fence();
            return false;
        };

        // ---- EXACT path for the decoded sub-chunk: record by record ----
        auto exact_subchunk = [&]() {
            n_slow++;
            n_records += (unsigned long long)ncons;
            // only the records that do something: a report about a hot subject, a DOWN report, or a batch end
            for (unsigned long long todo = wave_ballot((lane < ncons) & ((bits != 0u) | down | eob)); todo != 0ull;
                 todo &= todo - 1ull) {
                const int q = __ffsll((long long)todo) - 1;
                const int qdst = lane_value(dst, q);
                const unsigned int qbits = (unsigned)lane_value((int)bits, q);
                const int qflags = lane_value((int)down | ((int)eob << 1), q);
                if (qflags & 1) s.seen_down = true;  // R/MultiNodeCutDetector.java:89-91, hot subject or not
                exact_apply(d, s, pend, qdst, qbits, (qflags & 1) != 0, lane, nullptr, 0, nullptr);
                if (qflags & 2) {
                    exact_batch_end(d, s, pend, lane, &n_applied, &n_full);
                    if (s.batch_emitted) {  // R/MembershipService.java:333-335
                        emit_batch = s.batch;
                        break;
                    }
                    s.batch++;
                }
            }
            advance(ncons);
        };

        // ---- the record stream: LDS-DMA into this wave's ring, kDepth KiB always in flight ----
        // Stream KiB k (bytes [1024 k, 1024 k + 1024) from a0) lives in ring slot k % kRingSlots.  `landed` KiB have
        // arrived; KiB landed .. landed + kDepth - 1 are in flight, so one more has landed once at most kDepth - 1 loads
        // are outstanding.  KiB past the end of the stream are out of range of the buffer resource: they cost no memory
        // traffic but keep the count of outstanding loads constant, which keeps every wait a compile-time constant.
        int landed = 0, slot_issue = 0;  // KiB landed .. landed + kDepth - 1 are in flight; slot_issue = (landed + kDepth) % kRingSlots
#if RAPID_DMA_PAIRS
        lds_addr_t issue_lds = ring_lds;  // = ring_lds + slot_issue * kSlotBytes, kept as an address (no shift + add per load)
#endif
        auto stream_start = [&]() {
            landed = 0;
            slot_issue = kDepth % kRingSlots;
#if RAPID_DMA_PAIRS
            issue_lds = ring_lds + (kDepth % kRingSlots) * kSlotBytes;
#endif
            if (prestarted) {  // issued at the end of the previous receiver
                prestarted = false;
                return;
            }
            wait_dma<0>();  // an abandoned pass over this stream may still be landing
            wave_lds_fence();
            issue_head(cur);
        };
        // Makes the records [pos, end_rec) resident.  A slot is recycled only when every record in it has been
        // consumed: a window of <= kLeanWindow records touches <= kWindowSlots slots, so need <= kp + kWindowSlots (kp = the KiB
        // `pos` lies in) and no KiB up to landed + kDepth can reuse the slot of a KiB >= kp.
        auto stream_ensure = [&](int end_rec) {
            const int need = (int)((unsigned int)(delta + kRecBytes * end_rec + kSlotBytes - 1) / (unsigned int)kSlotBytes);
#if RAPID_DMA_PAIRS
            static_assert(kDepth >= 2, "pairs need two KiB in flight");
            // Two at a time: with at most kDepth - 2 loads outstanding two more KiB have landed.  The second load reuses the
            // slot of KiB landed - (kRingSlots - kDepth - 1), still older than the KiB `pos` lies in (landed + 1 < need).
            while (landed + 1 < need) {
                wait_dma<kDepth - 2>();
                lds_dma16(rsrc, lane16, (unsigned int)(landed + kDepth) * kSlotBytes, issue_lds);
                issue_lds += kSlotBytes;
                if (issue_lds == ring_lds + kRingBytes) issue_lds = ring_lds;
                lds_dma16(rsrc, lane16, (unsigned int)(landed + kDepth + 1) * kSlotBytes, issue_lds);
                issue_lds += kSlotBytes;
                if (issue_lds == ring_lds + kRingBytes) issue_lds = ring_lds;
                landed += 2;
            }
            while (landed < need) {
                wait_dma<kDepth - 1>();
                lds_dma16(rsrc, lane16, (unsigned int)(landed + kDepth) * kSlotBytes, issue_lds);
                issue_lds += kSlotBytes;
                if (issue_lds == ring_lds + kRingBytes) issue_lds = ring_lds;
                ++landed;
            }
#endif
            while (landed < need) {
#ifdef RAPID_TIMER_FINE
                RAPID_T0(te0);
#endif
                wait_dma<kDepth - 1>();
#ifdef RAPID_TIMER_FINE
                RAPID_T1(t_ensure, te0);
#endif
                lds_dma16(rsrc, lane16, (unsigned int)(landed + kDepth) * kSlotBytes, ring_lds + slot_issue * kSlotBytes);
                if (++slot_issue == kRingSlots) slot_issue = 0;
                ++landed;
            }
            wave_lds_fence();
        };

        bool from_careful = true;  // the lean path (re)establishes its witness on entry
        while (emit_batch < 0 && (restart || pos < nrec)) {
            if (restart) {
                // ---- detector state: nothing reported yet ----
                uint4* st = reinterpret_cast<uint4*>(mine);
                for (int i = lane; i < state_bytes / 16; i += kWave) st[i] = make_uint4(0, 0, 0, 0);
                s.running = 0;
                s.npend = 0;
                s.batch = 0;
                s.proposal_count = 0;
                s.seen_down = false;
                s.need_full = false;
                s.batch_emitted = false;
                pos = 0;
                ring_pos = delta;
                witness = -1;
#if RAPID_EARLY_CERT
                below_h = pot_below_h;
#endif
#if RAPID_FAST_WINDOW
                owed_reports = false;
                repicked = false;
#endif
                restart = false;
                careful_budget = 0;
                careful_cap = kWave;
                from_careful = false;  // nothing to recount: no slot has a report yet
                stream_start();
                continue;
            }
            if ((p.flags & 32) != 0) {  // measurement aid: stream the records through the ring without tallying them
                stream_ensure(min(pos + kLeanWindow, nrec));
                advance(min(kLeanWindow, nrec - pos));
                continue;
            }
            if (exact_only || s.batch_emitted || s.need_full || careful_budget > 0 || (p.flags & 8) != 0) {
                // ================= CAREFUL path: one sub-chunk of <= 64 records =================
                n_careful++;
#ifdef RAPID_TRACE
                if (lane == 0) fprintf(stderr, "C r=%d pos=%d run=%d npend=%d cap=%d budget=%d\n", r, pos, s.running, s.npend, careful_cap, careful_budget);
#endif
                from_careful = true;
#if RAPID_EARLY_CERT
                below_h = false;
#endif
#if RAPID_FAST_WINDOW && defined(RAPID_TRACE)
                if (owed_reports && lane == 0) fprintf(stderr, "BUG r=%d pos=%d: implicit reports owed on entry to the careful path\n", r, pos);
#endif
                stream_ensure(min(pos + kWave, nrec));
                RAPID_T0(tc0);
                decode();
#if RAPID_CAREFUL_HINT
                if (exact_only || s.batch_emitted || exact_next) {
                    exact_next = false;
                    exact_subchunk();
                    if (!exact_only && !s.batch_emitted) careful_cap = kWave;
#else
                if (exact_only || s.batch_emitted) {
                    exact_subchunk();
#endif
                } else if (immediate_subchunk()) {
                    careful_cap = min(kWave, careful_cap * 2);
                } else if (!restart) {
#if RAPID_CAREFUL_HINT
                    if (hint_cap >= 1 && hint_cap < ncons) {  // the records before the critical crossing
                        careful_cap = hint_cap;
                        RAPID_T1(t_careful, tc0);
                        continue;
                    }
                    if (hint_cap == 0 && ncons > 2) {  // the first record is the critical one
                        careful_cap = 2;
                        exact_next = true;
                        RAPID_T1(t_careful, tc0);
                        continue;
                    }
#endif
                    // An emission cannot be excluded somewhere in these ncons records.  Narrow the window instead of
                    // replaying all of them one by one: the record-by-record path only ever runs on a few records.
                    if (ncons > 4) {
                        careful_cap = ncons / 2;
                        RAPID_T1(t_careful, tc0);
                        continue;  // same position, smaller sub-chunk
                    }
                    exact_subchunk();
                    careful_cap = kWave;
                }
                if (careful_budget > 0) --careful_budget;
                RAPID_T1(t_careful, tc0);
                continue;
            }
            // ================= LEAN path: certified windows, implicit invalidation deferred =================
            if (from_careful) {  // the careful path keeps updatesInProgress itself; the lean path needs a witness
                from_careful = false;
                recount();
            }
            const int pos_in = pos;
            int gave_up = 0;
            while (nrec - pos >= kLeanWindow) {
                stream_ensure(pos + kLeanWindow);
                RAPID_T0(tl0);
#if RAPID_FAST_WINDOW
                int ok_;
                if (witness >= 0 && s.seen_down) {
                    // The first witness is whichever subject crossed L first; a few windows on there are many to choose
                    // from: one sweep picks the one with the most head room instead of waiting for the first to fail.
                    if (!repicked && witness_mask != 0u && pos >= kRepickAfter) {
                        repicked = true;
                        recount();
                    }
                    ok_ = fast_window();
                } else
#if RAPID_LEAN_V2
                    ok_ = s.seen_down ? lean_window(std::false_type{}, std::true_type{}) : lean_window(std::false_type{}, std::false_type{});
#else
                    ok_ = lean_window(std::false_type{}, std::false_type{});
#endif
#elif RAPID_LEAN_V2
                const int ok_ = s.seen_down ? lean_window(std::false_type{}, std::true_type{})
                                            : lean_window(std::false_type{}, std::false_type{});
#else
                const int ok_ = lean_window(std::false_type{}, std::false_type{});
#endif
                RAPID_T1(t_lean, tl0);
                if (!ok_) {
                    gave_up = 1;
                    break;
                }
                if (s.npend >= kPendCap / 2) {
                    RAPID_T0(tf0);
                    flush_pending();
                    RAPID_T1(t_flush, tf0);
                }
            }
            if (!gave_up && pos < nrec) {  // the tail: less than a window, the last record closes the last batch
                stream_ensure(nrec);
                if (!lean_window(std::true_type{}, std::false_type{})) gave_up = 1;
            }
            n_fast += (unsigned long long)((pos - pos_in + kLeanWindow - 1) / kLeanWindow);
            n_records += (unsigned long long)(pos - pos_in);
            if (gave_up) {
                // No certificate for the window at `pos`.  If the lean path got anywhere, `pos` is a batch end: apply what
                // is owed and look for a new witness -- with one, the lean path goes on.  Otherwise the careful path takes
                // the next sub-chunks (for longer and longer if the lean path keeps giving up at once).
                n_pipe++;
                RAPID_T0(tf1);
                const int old_witness = witness;
                if (pos > pos_in) flush_pending();  // never in the middle of a batch: the reference has not done it yet
                recount();
                RAPID_T1(t_flush, tf1);
                if (pos == pos_in || witness < 0 || witness == old_witness) {
                    careful_next = pos == pos_in ? min(careful_next * 2, 16) : 1;
                    careful_budget = careful_next;
                }
#if RAPID_CAREFUL_HINT
                // End game: most subjects are through and only a few are still between L and H.  Every witness the lean
                // path could pick is about to leave; each further attempt costs a rolled-back window and a sweep.
                if (running_exact && s.running <= kEndGameRunning && 2 * n_at_h >= d.n_scan) careful_budget = 1 << 20;
                // ... or the witness just chosen did not survive a single window while most subjects are through
                if (pos == pos_in && 2 * n_at_h >= d.n_scan) careful_budget = 1 << 20;
#endif
            }
        }
        wait_dma<0>();  // the ring is free: nothing of this stream is still landing
        RAPID_T0(to0);
        // start the next receiver's stream now; its first KiB land while this receiver's results are written
        Stream nxt = make_stream(0, 0);
        if (r_next < p.n_receivers) {
            nxt = make_stream(uniform64(n0_v), uniform64(n1_v));
            wave_lds_fence();
            issue_head(nxt);
            prestarted = true;
        }

        // ---- outputs: the proposal = every flushed (hot) slot, ascending node index ----
        int count = 0;
        unsigned long long fp = 0;
        if (emit_batch >= 0) {
            wave_lds_fence();
            int* const out = p.props + (long long)RAPID_RX(r) * p.prop_cap;
            for (int i0 = 0; i0 < d.n_scan; i0 += kWave) {
                const int i = i0 + lane;
                const bool take = i < d.n_scan && (d.load(i) & kFlushed) != 0;
                const unsigned long long mk = wave_ballot(take);
                const int idx = count + __popcll(mk & lanes_lt(lane));
                if (take) {
                    const int node = node_of_slot[i];
                    if (idx < p.prop_cap) out[idx] = node;
                    fp += mix64((unsigned long long)node);
                }
                count += __popcll(mk);
            }
            fp = wave_sum64(fp) + mix64(0x5EEDull + (unsigned long long)count);
            if (fp == 0) fp = 1;
        }
        if (lane == 0) {
            p.emit_batch[RAPID_RX(r)] = emit_batch;
            p.num_proposals[RAPID_RX(r)] = s.proposal_count;
            p.prop_count[RAPID_RX(r)] = count > p.prop_cap ? -1 : count;
#ifdef RAPID_PHASE_TIMERS
            // profiling build: cycles spent on this receiver, in total and per phase
#ifdef RAPID_TIMER_REALTIME
            // shader cycles | 10-ns ticks of the constant-rate counter << 32 | start tick (low 24 bits) << 40 is not needed:
            p.fingerprint[RAPID_RX(r)] = ((__builtin_amdgcn_s_memtime() - t_rx0) & 0xFFFFFFFFull) |
                                         ((__builtin_amdgcn_s_memrealtime() - t_real0) << 32);
            p.emit_batch[RAPID_RX(r)] = (int)(t_real0 & 0x7FFFFFFFull);  // when the receiver was started
#else
            p.fingerprint[RAPID_RX(r)] = ((__builtin_amdgcn_s_memtime() - t_rx0) & 0xFFFFFFFFull) | ((t_ensure - t_ensure0) << 32);
#endif
            p.num_proposals[RAPID_RX(r)] = (int)((t_careful - t_careful0) >> 4);
            p.prop_count[RAPID_RX(r)] = (int)((t_lean - t_lean0) >> 4);
#else
            p.fingerprint[RAPID_RX(r)] = fp;
#endif
        }
        wave_lds_fence();
        r = r_next;
        cur = nxt;
        RAPID_T1(t_out, to0);
#ifdef RAPID_PHASE_TIMERS
        t_rx++;
#endif
    }
    unsigned long long mine_stats[8];
#ifdef RAPID_PHASE_TIMERS
    RAPID_T1(t_total, t_kernel0);
    mine_stats[0] = t_total; mine_stats[1] = t_ensure; mine_stats[2] = t_lean; mine_stats[3] = t_careful;
    mine_stats[4] = t_out; mine_stats[5] = t_flush; mine_stats[6] = t_rx; mine_stats[7] = n_fast;
#else
    mine_stats[0] = n_slow; mine_stats[1] = n_fast; mine_stats[2] = (unsigned long long)n_full; mine_stats[3] = n_restart;
    mine_stats[4] = (unsigned long long)n_applied; mine_stats[5] = n_records; mine_stats[6] = n_pipe; mine_stats[7] = n_careful;
#endif
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) atomicAdd(&block_stats[i], mine_stats[i]);
    }
    __syncthreads();
    // p.stats = [gridDim.x][8], accumulated over launches; one plain read-modify-write per workgroup and counter
    if (threadIdx.x < 8u && p.stats != nullptr) p.stats[(size_t)blockIdx.x * 8 + threadIdx.x] += block_stats[threadIdx.x];
}

// --------------------------------------------------------------------------------------------------------------
// One MultiNodeCutDetector instance with its state in global memory (parity API rapid_cd_*): exact path only.
// scal[0]=updatesInProgress [1]=proposalCount [2]=seenLinkDownEvents.  mode 0: apply n alerts (no filter --
// the class itself has none); mode 1: invalidateFailingEdges (literal full pass); grid = 1, block = 64.
// --------------------------------------------------------------------------------------------------------------
struct CdParams {
    unsigned short* state;        // [n_nodes rounded to even] node-indexed masks
    int* scal;                    // [4]
    const unsigned char* alerts;  // packed records (mode 0)
    int n_alerts;
    int n_nodes, K, H, L;
    const int* obs;   // view's observer table (mode 1)
    int* out_idx;     // concatenated emissions
    int out_cap;
    int* out_counts;  // [n_alerts] (mode 0) / [1] (mode 1)
    int* out_n;       // total emitted (may exceed out_cap -> caller reports ECAPACITY)
    int mode;
};

__global__ __launch_bounds__(64) void cd_instance_kernel(CdParams p) {
    const int lane = (int)threadIdx.x;
    TableDetector d;
    d.st16 = p.state;
    d.st32 = reinterpret_cast<unsigned int*>(p.state);
    d.obs = p.obs;
    d.n_scan = p.n_nodes;
    d.K = p.K;
    d.H = p.H;
    d.L = p.L;
    d.kmask = (1u << p.K) - 1u;
    RxScalars s;
    s.running = p.scal[0];
    s.proposal_count = p.scal[1];
    s.seen_down = p.scal[2] != 0;
    s.npend = 0;
    s.batch = 0;
    s.need_full = true;
    s.batch_emitted = false;
    int total = 0;
    if (p.mode == 0) {
        for (int a = 0; a < p.n_alerts; ++a) {
            const unsigned int* w = reinterpret_cast<const unsigned int*>(p.alerts + (long long)a * kRecBytes);
            const int dst = uniform((int)w[3]);
            const unsigned int w4 = uniform(w[4]);
            const int before = total;
            if ((unsigned)dst < (unsigned)p.n_nodes)
                exact_apply(d, s, nullptr, dst, w4 & d.kmask, ((w4 >> 16) & 0xFFu) != 0, lane, p.out_idx, p.out_cap,
                            &total);
            if (lane == 0) p.out_counts[a] = total - before;
        }
    } else if (s.seen_down) {  // R/MultiNodeCutDetector.java:140-142
        const int nH = invalidate_table(d, lane);
        s.running -= nH;
        if (nH > 0 && s.running == 0) {
            s.proposal_count++;
            flush_sweep(d, lane, p.out_idx, p.out_cap, &total);
        }
    }
    if (p.mode != 0 && lane == 0) p.out_counts[0] = total;
    d.sync();
    if (lane == 0) {
        p.scal[0] = s.running;
        p.scal[1] = s.proposal_count;
        p.scal[2] = s.seen_down ? 1 : 0;
        *p.out_n = total;
    }
}

// --------------------------------------------------------------------------------------------------------------
// Measurement probe (not part of the product path): the same access pattern as the tally kernel -- one wave per
// receiver stream, TILE-byte tiles of 16 B/lane loads, DEPTH tiles in flight -- with no processing, to separate
// what the memory system delivers for this pattern from what the detector logic costs.
// --------------------------------------------------------------------------------------------------------------
template <int TILE_VEC, int DEPTH>
__global__ __launch_bounds__(1024) void stream_probe_kernel(const unsigned char* records, unsigned long long records_bytes,
                                                            const long long* rec_off, int n_receivers,
                                                            unsigned int* next_receiver, unsigned int* sink) {
    const int lane = (int)(threadIdx.x & 63u);
    unsigned int acc = 0;
    for (;;) {
        int r = 0;
        if (lane == 0) r = (int)atomicAdd(next_receiver, 1u);
        r = __builtin_amdgcn_readfirstlane(r);
        if (r >= n_receivers) break;
        const unsigned long long b0 = (unsigned long long)rec_off[r] * 20ull, b1 = (unsigned long long)rec_off[r + 1] * 20ull;
        const unsigned long long a0 = b0 & ~15ull;
        const int tile_bytes = TILE_VEC * 1024;
        const int ntiles = (int)((b1 - a0 + tile_bytes - 1) / tile_bytes);
        const unsigned long long last16 = records_bytes - 16ull;
        uint4 t[DEPTH][TILE_VEC];
#pragma unroll
        for (int q = 0; q < DEPTH; ++q)
#pragma unroll
            for (int m = 0; m < TILE_VEC; ++m) {
                unsigned long long addr = a0 + (unsigned long long)q * tile_bytes + 16ull * (unsigned)(lane + 64 * m);
                addr = addr < last16 ? addr : last16;
                t[q][m] = *reinterpret_cast<const uint4*>(records + addr);
            }
        for (int jb = 0; jb < ntiles; jb += DEPTH) {
#pragma unroll
            for (int q = 0; q < DEPTH; ++q) {
#pragma unroll
                for (int m = 0; m < TILE_VEC; ++m) {
                    acc ^= t[q][m].x ^ t[q][m].y ^ t[q][m].z ^ t[q][m].w;
                    const int j = jb + q + DEPTH;
                    unsigned long long addr = a0 + (unsigned long long)(j < ntiles ? j : 0) * tile_bytes + 16ull * (unsigned)(lane + 64 * m);
                    addr = addr < last16 ? addr : last16;
                    t[q][m] = *reinterpret_cast<const uint4*>(records + addr);
                }
            }
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

// The same probe through the LDS-DMA path: kD KiB in flight per wave, landing in a private LDS ring of kD slots.
template <int kD>
__global__ __launch_bounds__(1024) void dma_probe_kernel(const unsigned char* records, unsigned long long records_bytes,
                                                         const long long* rec_off, int n_receivers,
                                                         unsigned int* next_receiver, unsigned int* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = (int)(threadIdx.x & 63u);
    const int wave = (int)(threadIdx.x >> 6);
    // sink[1] = byte offset of the rings inside the workgroup's LDS allocation (measurement knob)
    const lds_addr_t ring = lds_uniform(lds_address(smem + sink[1] + wave * kD * 1024));
    const unsigned int lane16 = (unsigned int)lane * 16u;
    for (;;) {
        int r = 0;
        if (lane == 0) r = (int)atomicAdd(next_receiver, 1u);
        r = __builtin_amdgcn_readfirstlane(r);
        if (r >= n_receivers) break;
        const unsigned long long b0 = (unsigned long long)rec_off[r] * 20ull, b1 = (unsigned long long)rec_off[r + 1] * 20ull;
        const unsigned long long a0 = b0 & ~15ull;
        const int nk = (int)((b1 - a0 + 1023ull) / 1024ull);
        const dma_rsrc_t rsrc = dma_make_rsrc(records + a0, (unsigned int)(((b1 - a0) + 15ull) & ~15ull));
        wait_dma<0>();
#pragma unroll
        for (int k = 0; k < kD; ++k) lds_dma16(rsrc, lane16, (unsigned int)k * 1024u, ring + k * 1024);
        int slot = 0;
        for (int k = 0; k < nk; ++k) {
            wait_dma<kD - 1>();
            lds_dma16(rsrc, lane16, (unsigned int)(k + kD) * 1024u, ring + slot * 1024);
            if (++slot == kD) slot = 0;
        }
    }
    wait_dma<0>();
    if (reinterpret_cast<unsigned int*>(smem)[threadIdx.x] == 0x12345678u) sink[0] = 1u;
}

}  // namespace rapid
