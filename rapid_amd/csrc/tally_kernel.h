// Alert-tally kernels for gfx950 (CDNA4, wave64) -- the hot path.
//
// What they compute (reference semantics, R/ = /root/reference/rapid/src/main/java/com/vrg/rapid/):
//   per receiver: MembershipService.handleMessage(BatchedAlertMessage)  R/MembershipService.java:300-354
//                 + filterAlertMessages                                 R/MembershipService.java:644-675
//                 + MultiNodeCutDetector.aggregateForProposal           R/MultiNodeCutDetector.java:76-128
//                 + MultiNodeCutDetector.invalidateFailingEdges         R/MultiNodeCutDetector.java:137-164
//
// Mapping onto the machine (DESIGN.md "Tally kernel"):
//   * one wavefront (= one 64-thread workgroup) per simulated receiver; the receiver's whole detector state
//     is a 16-bit word per node in LDS: bits 0..K-1 = rings reported, bit 14 = already flushed into an
//     emitted proposal, bit 15 = node is a member of the current view;
//   * the delivered stream is read ONCE from HBM in 4 KiB tiles with 16 B/lane coalesced loads (next tile in
//     flight in registers while the current one is consumed), staged through an LDS ring, and consumed in
//     sub-chunks of up to 64 records (one per lane) that always end at a batch end when they contain one;
//   * FAST path per 64-record sub-chunk: order-free ds_or_rtn on the masks; the L/H watermark crossings of
//     the sub-chunk are counted with wave ballots; implicit edge invalidation is applied once per sub-chunk,
//     only for the nodes that crossed L (incremental form), as of the last batch end in the sub-chunk;
//   * the reference's detector is a sequential state machine: an emission can only happen on an H crossing
//     that finds updatesInProgress == 0.  If (updatesInProgress before the sub-chunk) - (H crossings in it)
//     >= 1, no emission is possible under ANY order and the order-free result is exact; otherwise the
//     sub-chunk is rolled back (undo = clearing exactly the bits each lane set) and replayed by the EXACT
//     path, record by record, with the implicit invalidation after every batch end;
//   * after the batch that announces a proposal the receiver ignores the rest of its stream
//     (announcedProposal, R/MembershipService.java:318-319) -- the wave stops reading.
//
// No MFMA: the path is integer scatter/popcount.  No static __shared__: all LDS is carved from the
// 16-byte aligned dynamic segment.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rapid {

constexpr int kWave = 64;
constexpr int kRecBytes = 20;
constexpr int kTileBytes = 4096;                     // one tile = 4 x (64 lanes x 16 B)
constexpr int kRingTiles = 2;                        // LDS ring of tiles the sub-chunks are read from
constexpr int kRingBytes = kTileBytes * kRingTiles;  // power of two
constexpr int kStageBytes = kRingBytes;
constexpr int kPendCap = 128;                        // nodes that crossed L and still await invalidation
constexpr int kUndoCap = 128;                        // implicit bits set inside one sub-chunk
constexpr uint32_t kFlushed = 1u << 14;
constexpr uint32_t kMember = 1u << 15;

struct TallyParams {
    const unsigned char* records;    // packed rapid_alert_record[]
    unsigned long long records_bytes;  // readable bytes at `records` (>= 16 past the last record)
    const long long* rec_off;        // [R+1], in records
    int n_receivers;
    int n_nodes;
    int K, H, L;
    long long cfg_id;
    const unsigned short* state_template;  // [n_nodes rounded up to 8] kMember set for members
    const int* obs;                        // [n_nodes][K] observers (expected observers for non-members)
    const int* subj;                       // [n_nodes][K] subjects (-1 rows for non-members)
    int* emit_batch;                       // [R]
    int* num_proposals;                    // [R]
    int* prop_count;                       // [R]; -1 if the proposal did not fit prop_cap
    unsigned long long* fingerprint;       // [R]
    int* props;                            // [R][prop_cap] ascending node index
    int prop_cap;
    unsigned long long* stats;             // [8]
    int force_exact;
};

__host__ __device__ inline int tally_state_bytes(int n_nodes) { return ((n_nodes * 2 + 15) / 16) * 16; }
__host__ __device__ inline int tally_lds_bytes(int n_nodes) {
    return tally_state_bytes(n_nodes) + kStageBytes + kPendCap * 4 + kUndoCap * 4;
}

// ---- small wave helpers (block == one wave of 64) ------------------------------------------------------
__device__ __forceinline__ unsigned long long lanes_lt(int lane) { return (1ull << lane) - 1ull; }
__device__ __forceinline__ int wave_sum(int v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, kWave);
    return v;
}
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

// Per-receiver detector state.  `st` may point to LDS (population kernel) or to global memory (single
// MultiNodeCutDetector instance); all accesses go through 16-bit loads/stores and 32-bit atomics.
struct Detector {
    unsigned short* st16;
    unsigned int* st32;
    const int* obs;
    const int* subj;
    int n_nodes, K, H, L;
    unsigned int kmask;

    __device__ __forceinline__ unsigned int load(int n) const { return st16[n]; }
    __device__ __forceinline__ int count(unsigned int m) const { return __popc(m & kmask); }
    // atomically OR `bits` into node n's mask; returns the previous 16-bit word
    __device__ __forceinline__ unsigned int or_bits(int n, unsigned int bits) const {
        const int sh = (n & 1) * 16;
        const unsigned int old = atomicOr(&st32[n >> 1], bits << sh);
        return (old >> sh) & 0xFFFFu;
    }
    __device__ __forceinline__ void clear_bits(int n, unsigned int bits) const {
        const int sh = (n & 1) * 16;
        atomicAnd(&st32[n >> 1], ~(bits << sh));
    }
};

// Implicit-edge invalidation (R/MultiNodeCutDetector.java:137-164), incremental form: only the nodes in
// pend[0..n_elig) -- those that crossed L since the previous pass -- can enable a new (observer, subject)
// pair.  Each lane handles one (entrant, role, ring) triple.  Returns the number of H crossings caused.
// If undo != nullptr every bit actually set is appended to undo[] (node | ring << 24); *n_undo counts them
// even beyond kUndoCap so the caller can detect overflow.
__device__ inline int invalidate_entrants(const Detector& d, const unsigned int* pend, int n_elig, unsigned int* undo,
                                          int* n_undo, int lane, int* n_applied) {
    int nH = 0;
    const int twoK = 2 * d.K;
    const int total = n_elig * twoK;
    for (int q0 = 0; q0 < total; q0 += kWave) {
        const int q = q0 + lane;
        const bool active = q < total;
        int s = -1, o = -1, k = 0;
        if (active) {
            const int e = q / twoK;
            const int j = q - e * twoK;
            const int n = (int)pend[e];
            if (j < d.K) {  // entrant as the node in flux: its observers
                k = j;
                s = n;
                o = d.obs[n * d.K + k];
            } else {  // entrant as an observer: its subjects
                k = j - d.K;
                o = n;
                s = d.subj[n * d.K + k];
            }
        }
        const bool ok = active && s >= 0 && o >= 0;
        const unsigned int ms = ok ? d.load(s) : 0u;
        const unsigned int mo = ok ? d.load(o) : 0u;
        const int cs = d.count(ms), co = d.count(mo);
        const bool apply = ok && cs >= d.L && cs < d.H && co >= d.L && !(mo & kFlushed) && !(ms & (1u << k));
        unsigned int old = 0;
        if (apply) old = d.or_bits(s, 1u << k);
        const bool isnew = apply && !(old & (1u << k));
        const bool crossH = isnew && d.count(old) == d.H - 1;
        nH += __popcll(__ballot(crossH));
        const unsigned long long mnew = __ballot(isnew);
        if (undo != nullptr) {
            const int idx = *n_undo + __popcll(mnew & lanes_lt(lane));
            if (isnew && idx < kUndoCap) undo[idx] = (unsigned)s | ((unsigned)k << 24);
            *n_undo += __popcll(mnew);
        }
        *n_applied += __popcll(mnew);
        __syncthreads();
    }
    return nH;
}

// The reference's literal full pass: every node in preProposal x its K observers.  Used when the pending
// list overflowed or when joiners are in flux (their expected observers are not in the subjects table).
__device__ inline int invalidate_full(const Detector& d, unsigned int* undo, int* n_undo, int lane, int* n_applied) {
    int nH = 0;
    for (int n0 = 0; n0 < d.n_nodes; n0 += kWave) {
        const int n = n0 + lane;
        const unsigned int m = n < d.n_nodes ? d.load(n) : 0u;
        const int c = d.count(m);
        const bool inpre = n < d.n_nodes && c >= d.L && c < d.H;
        if (__ballot(inpre) == 0ull) continue;
        for (int k = 0; k < d.K; ++k) {
            const int o = inpre ? d.obs[n * d.K + k] : -1;
            const unsigned int mo = o >= 0 ? d.load(o) : 0u;
            const bool apply = o >= 0 && d.count(mo) >= d.L && !(mo & kFlushed) && !(m & (1u << k));
            unsigned int old = 0;
            if (apply) old = d.or_bits(n, 1u << k);
            const bool isnew = apply && !(old & (1u << k));
            const bool crossH = isnew && d.count(old) == d.H - 1;
            nH += __popcll(__ballot(crossH));
            const unsigned long long mnew = __ballot(isnew);
            if (undo != nullptr) {
                const int idx = *n_undo + __popcll(mnew & lanes_lt(lane));
                if (isnew && idx < kUndoCap) undo[idx] = (unsigned)n | ((unsigned)k << 24);
                *n_undo += __popcll(mnew);
            }
            *n_applied += __popcll(mnew);
        }
        __syncthreads();
    }
    return nH;
}

// An emission (R/MultiNodeCutDetector.java:116-123): every node that crossed H and was not yet returned is
// returned now and leaves `proposal`.  Marks them flushed; optionally appends them (ascending) to out[].
__device__ inline void flush_sweep(const Detector& d, int lane, int* out, int out_cap, int* out_n) {
    __syncthreads();
    for (int n0 = 0; n0 < d.n_nodes; n0 += kWave) {
        const int n = n0 + lane;
        const unsigned int m = n < d.n_nodes ? d.load(n) : 0u;
        const bool take = n < d.n_nodes && d.count(m) >= d.H && !(m & kFlushed);
        if (take) d.st16[n] = (unsigned short)(m | kFlushed);
        if (out_n != nullptr) {
            const unsigned long long mk = __ballot(take);
            const int idx = *out_n + __popcll(mk & lanes_lt(lane));
            if (take && out != nullptr && idx < out_cap) out[idx] = n;
            *out_n += __popcll(mk);
        }
    }
    __syncthreads();
}

// Scalars of one receiver (wave-uniform).
struct RxScalars {
    int running;        // updatesInProgress
    int npend;          // entries in pend[]
    int batch;          // batches fully processed
    int proposal_count; // getNumProposals()
    bool seen_down;     // seenLinkDownEvents
    bool need_full;     // incremental list no longer complete -> use the literal full pass
    bool batch_emitted; // some emission happened in the batch being processed
};

// EXACT application of one alert (all rings, ascending) -- R/MultiNodeCutDetector.java:76-128.  Executed
// redundantly by all lanes on wave-uniform values; lane 0 performs the stores.  Emissions go to the flush
// sweep; if emit_out != nullptr the returned nodes are appended there.
__device__ inline void exact_apply(const Detector& d, RxScalars& s, unsigned int* pend, int dst, unsigned int bits,
                                   bool down, int lane, int* emit_out, int emit_cap, int* emit_n) {
    if (bits == 0) return;
    if (down) s.seen_down = true;
    unsigned int m = d.load(dst);
    unsigned int nb = bits & ~m & d.kmask;
    while (nb) {
        const int k = __ffs((int)nb) - 1;
        nb &= nb - 1;
        m |= 1u << k;
        const int c = d.count(m);
        if (c == d.L) {
            s.running++;
            if (s.npend < kPendCap) {
                if (lane == 0) pend[s.npend] = (unsigned)dst;
                s.npend++;
            } else {
                s.need_full = true;
            }
        }
        if (c == d.H) {
            s.running--;
            if (s.running == 0) {
                s.proposal_count++;
                s.batch_emitted = true;
                __syncthreads();
                if (lane == 0) d.st16[dst] = (unsigned short)m;
                flush_sweep(d, lane, emit_out, emit_cap, emit_n);
                m = d.load(dst);
            }
        }
    }
    __syncthreads();
    if (lane == 0) d.st16[dst] = (unsigned short)m;
    __syncthreads();
}

// EXACT end-of-batch step: invalidateFailingEdges (R/MultiNodeCutDetector.java:137-164) as invoked at
// R/MembershipService.java:330.
__device__ inline void exact_batch_end(const Detector& d, RxScalars& s, unsigned int* pend, int lane, int* emit_out,
                                       int emit_cap, int* emit_n, int* n_applied, int* n_full) {
    if (!s.seen_down) return;
    int nH;
    if (s.need_full) {
        nH = invalidate_full(d, nullptr, nullptr, lane, n_applied);
        (*n_full)++;
    } else {
        nH = invalidate_entrants(d, pend, s.npend, nullptr, nullptr, lane, n_applied);
    }
    s.npend = 0;
    s.running -= nH;
    if (nH > 0 && s.running == 0) {
        s.proposal_count++;
        s.batch_emitted = true;
        flush_sweep(d, lane, emit_out, emit_cap, emit_n);
    }
}

// --------------------------------------------------------------------------------------------------------
// Whole-population tally: grid = receivers, block = 64.
// --------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void tally_population_kernel(TallyParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = (int)threadIdx.x;
    const int r = (int)blockIdx.x;
    if (r >= p.n_receivers) return;

    const int state_bytes = tally_state_bytes(p.n_nodes);
    unsigned char* const stage = smem + state_bytes;
    unsigned int* const pend = reinterpret_cast<unsigned int*>(stage + kStageBytes);
    unsigned int* const undo = pend + kPendCap;

    Detector d;
    d.st16 = reinterpret_cast<unsigned short*>(smem);
    d.st32 = reinterpret_cast<unsigned int*>(smem);
    d.obs = p.obs;
    d.subj = p.subj;
    d.n_nodes = p.n_nodes;
    d.K = p.K;
    d.H = p.H;
    d.L = p.L;
    d.kmask = (1u << p.K) - 1u;

    const long long rec0 = p.rec_off[r];
    const int nrec = (int)(p.rec_off[r + 1] - rec0);
    const unsigned long long b0 = (unsigned long long)rec0 * kRecBytes;
    const unsigned long long a0 = b0 & ~15ull;  // 16-B aligned start of this receiver's byte range
    const int delta = (int)(b0 - a0);
    const int ntiles = (int)(((long long)delta + (long long)nrec * kRecBytes + kTileBytes - 1) / kTileBytes);
    const unsigned int cfg_lo = (unsigned int)(unsigned long long)p.cfg_id;
    const unsigned int cfg_hi = (unsigned int)((unsigned long long)p.cfg_id >> 32);
    const unsigned int* const ring32 = reinterpret_cast<const unsigned int*>(stage);

    unsigned long long n_slow = 0, n_fast = 0, n_restart = 0, n_records = 0;
    int n_applied = 0, n_full = 0;
    int emit_batch = -1;
    RxScalars s;
    bool exact_only = p.force_exact != 0;

    for (int attempt = 0; attempt < 2; ++attempt) {
        // ---- state init: member bits from the template (L2-resident), everything else zero ----
        {
            const uint4* tpl = reinterpret_cast<const uint4*>(p.state_template);
            uint4* dst = reinterpret_cast<uint4*>(smem);
            for (int i = lane; i < state_bytes / 16; i += kWave) dst[i] = tpl[i];
        }
        s.running = 0;
        s.npend = 0;
        s.batch = 0;
        s.proposal_count = 0;
        s.seen_down = false;
        s.need_full = false;
        s.batch_emitted = false;
        emit_batch = -1;
        bool restart = false;

        // ---- tile pipeline: the stream is a ring of kRingTiles x 4 KiB in LDS; the next tile is in flight in
        // registers (16 B/lane coalesced loads) while sub-chunks are consumed from the ring ----
        uint4 tile[kTileBytes / 1024];
        int loaded = 0;  // tiles copied into the ring so far
        auto issue = [&](int j) {
            const unsigned long long g = a0 + (unsigned long long)j * kTileBytes;
#pragma unroll
            for (int m = 0; m < kTileBytes / 1024; ++m) {
                const unsigned long long addr = g + 16ull * (unsigned)(lane + kWave * m);
                tile[m] = (addr + 16 <= p.records_bytes) ? *reinterpret_cast<const uint4*>(p.records + addr)
                                                         : make_uint4(0, 0, 0, 0);
            }
        };
        if (ntiles > 0) issue(0);

        int pos = 0;          // next unconsumed record
        int pos_off = delta;  // (delta + 20 * pos) mod ring size
        while (pos < nrec && emit_batch < 0 && !restart) {
            const int navail = min(kWave, nrec - pos);
            const int need = (delta + kRecBytes * (pos + navail) - 1) / kTileBytes;
            while (loaded <= need) {
                __syncthreads();  // nobody still reads the slot being overwritten
                uint4* slot = reinterpret_cast<uint4*>(stage + (loaded % kRingTiles) * kTileBytes);
#pragma unroll
                for (int m = 0; m < kTileBytes / 1024; ++m) slot[lane + kWave * m] = tile[m];
                ++loaded;
                if (loaded < ntiles) issue(loaded);
                __syncthreads();
            }

            // ---- one record per lane ----
            const int off = (pos_off + kRecBytes * lane) & (kRingBytes - 1);
            const unsigned int w0 = ring32[off >> 2];
            const unsigned int w1 = ring32[((off + 4) & (kRingBytes - 1)) >> 2];
            const unsigned int w3 = ring32[((off + 12) & (kRingBytes - 1)) >> 2];
            const unsigned int w4 = ring32[((off + 16) & (kRingBytes - 1)) >> 2];
            const int dst = (int)w3;
            const bool down = ((w4 >> 16) & 0xFFu) != 0;
            bool eob = lane < navail && ((((w4 >> 24) & 1u) != 0) || pos + lane == nrec - 1);
            // a sub-chunk ends at its last batch end (if it has one): no record is applied before the batch
            // end that precedes it has been processed
            const unsigned long long mE_all = __ballot(eob);
            const int lastE = mE_all ? 63 - __clzll((long long)mE_all) : -1;
            const int ncons = lastE >= 0 ? lastE + 1 : navail;
            const bool valid = lane < ncons;
            eob = eob && valid;
            // filterAlertMessages (R/MembershipService.java:644-675)
            bool pass = valid && w0 == cfg_lo && w1 == cfg_hi && (unsigned)dst < (unsigned)p.n_nodes;
            const unsigned int m0 = pass ? d.load(dst) : 0u;
            pass = pass && (((m0 & kMember) != 0) == down);
            const unsigned int bits = pass ? (w4 & d.kmask) : 0u;
            n_records += (unsigned long long)ncons;

            // once an emission happened inside the current batch, the rest of that batch (whose end announces
            // the proposal) is processed exactly
            bool replay = exact_only || s.batch_emitted;
            if (!replay) {
                // ---------------- FAST path: order-free, then a safety check ----------------
                unsigned int old = 0;
                if (bits) old = d.or_bits(dst, bits);
                const unsigned int newbits = bits & ~old;
                const int c0 = d.count(old), c1 = d.count(old | bits);
                const bool isL = bits != 0 && c0 < d.L && c1 >= d.L;
                const bool isH = bits != 0 && c0 < d.H && c1 >= d.H;
                const unsigned long long mL = __ballot(isL), mH = __ballot(isH);
                const unsigned long long mD = __ballot(bits != 0 && down);
                const unsigned long long mJ = __ballot(bits != 0 && !down);  // joiner (UP) reports
                const int nLc = __popcll(mL), nHc = __popcll(mH);
                const bool seen = s.seen_down || mD != 0ull;
                bool need_full = s.need_full || mJ != 0ull;
                // append the nodes that crossed L (lane order)
                const int posn = s.npend + __popcll(mL & lanes_lt(lane));
                if (isL && posn < kPendCap) pend[posn] = (unsigned)dst;
                const int npend_new = s.npend + nLc;
                if (npend_new > kPendCap) need_full = true;
                const bool run_inv = lastE >= 0 && seen;
                __syncthreads();
                int nHi = 0, n_undo = 0, applied_here = 0;
                if (run_inv) {
                    if (need_full) {
                        nHi = invalidate_full(d, undo, &n_undo, lane, &applied_here);
                        n_full++;
                    } else {
                        nHi = invalidate_entrants(d, pend, npend_new, undo, &n_undo, lane, &applied_here);
                    }
                }
                const int Htot = nHc + nHi;
                // No emission is possible inside this sub-chunk under ANY order if updatesInProgress cannot
                // reach 0 at one of its H crossings.
                const bool safe = Htot == 0 || s.running - Htot >= 1;
                if (safe) {
                    s.running += nLc - Htot;
                    s.seen_down = seen;
                    s.batch += __popcll(mE_all & ((ncons == 64) ? ~0ull : ((1ull << ncons) - 1ull)));
                    s.need_full = need_full;
                    s.npend = (run_inv || need_full) ? 0 : npend_new;
                    n_applied += applied_here;
                    n_fast++;
                } else if (n_undo > kUndoCap) {
                    restart = true;  // cannot roll back: redo this receiver on the exact path only
                } else {
                    // ---------------- roll back, then replay exactly ----------------
                    __syncthreads();
                    for (int u = lane; u < n_undo; u += kWave) {
                        const unsigned int e = undo[u];
                        d.clear_bits((int)(e & 0xFFFFFFu), 1u << (e >> 24));
                    }
                    if (newbits) d.clear_bits(dst, newbits);
                    __syncthreads();
                    replay = true;
                }
            }
            if (replay && !restart) {
                // ---------------- EXACT path: record by record ----------------
                n_slow++;
                for (int q = 0; q < ncons; ++q) {
                    const int qdst = __shfl(dst, q, kWave);
                    const unsigned int qbits = (unsigned)__shfl((int)bits, q, kWave);
                    const int qflags = __shfl((int)down | ((int)eob << 1), q, kWave);
                    if (qbits != 0 && !(qflags & 1)) s.need_full = true;  // joiner report
                    exact_apply(d, s, pend, qdst, qbits, (qflags & 1) != 0, lane, nullptr, 0, nullptr);
                    if (qflags & 2) {
                        exact_batch_end(d, s, pend, lane, nullptr, 0, nullptr, &n_applied, &n_full);
                        if (s.batch_emitted) {  // R/MembershipService.java:333-335
                            emit_batch = s.batch;
                            break;
                        }
                        s.batch++;
                    }
                }
            }
            pos += ncons;
            pos_off = (pos_off + kRecBytes * ncons) & (kRingBytes - 1);
        }
        if (!restart) break;
        exact_only = true;
        n_restart++;
        __syncthreads();
    }

    // ---- outputs ----
    int count = 0;
    unsigned long long fp = 0;
    if (emit_batch >= 0) {
        int* const out = p.props + (long long)r * p.prop_cap;
        for (int n0 = 0; n0 < p.n_nodes; n0 += kWave) {
            const int n = n0 + lane;
            const bool take = n < p.n_nodes && (d.load(n) & kFlushed) != 0;
            const unsigned long long mk = __ballot(take);
            const int idx = count + __popcll(mk & lanes_lt(lane));
            if (take) {
                if (idx < p.prop_cap) out[idx] = n;
                fp += mix64((unsigned long long)n);
            }
            count += __popcll(mk);
        }
        fp = wave_sum64(fp) + mix64(0x5EEDull + (unsigned long long)count);
        if (fp == 0) fp = 1;
    }
    if (lane == 0) {
        p.emit_batch[r] = emit_batch;
        p.num_proposals[r] = s.proposal_count;
        p.prop_count[r] = count > p.prop_cap ? -1 : count;
        p.fingerprint[r] = fp;
        if (p.stats != nullptr) {
            atomicAdd(&p.stats[0], n_slow);
            atomicAdd(&p.stats[1], n_fast);
            atomicAdd(&p.stats[2], (unsigned long long)n_full);
            atomicAdd(&p.stats[3], n_restart);
            atomicAdd(&p.stats[4], (unsigned long long)n_applied);
            atomicAdd(&p.stats[5], n_records);
        }
    }
}

// --------------------------------------------------------------------------------------------------------
// One MultiNodeCutDetector instance with its state in global memory (parity API rapid_cd_*): exact path only.
// scal[0]=updatesInProgress [1]=proposalCount [2]=seenLinkDownEvents.  mode 0: apply n alerts (no filter --
// the class itself has none); mode 1: invalidateFailingEdges (literal full pass); grid = 1, block = 64.
// --------------------------------------------------------------------------------------------------------
struct CdParams {
    unsigned short* state;  // [n_nodes rounded to even], bit15 = member (refreshed by the host from the view)
    int* scal;              // [4]
    const unsigned char* alerts;  // packed records (mode 0)
    int n_alerts;
    int n_nodes, K, H, L;
    const int* obs;
    const int* subj;
    int* out_idx;     // concatenated emissions
    int out_cap;
    int* out_counts;  // [n_alerts] (mode 0) / [1] (mode 1)
    int* out_n;       // total emitted (may exceed out_cap -> caller reports ECAPACITY)
    int mode;
};

__global__ __launch_bounds__(64) void cd_instance_kernel(CdParams p) {
    const int lane = (int)threadIdx.x;
    Detector d;
    d.st16 = p.state;
    d.st32 = reinterpret_cast<unsigned int*>(p.state);
    d.obs = p.obs;
    d.subj = p.subj;
    d.n_nodes = p.n_nodes;
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
    s.need_full = true;  // the instance API always uses the literal full pass
    s.batch_emitted = false;
    unsigned int dummy_pend = 0;
    int total = 0, n_applied = 0, n_full = 0;
    if (p.mode == 0) {
        for (int a = 0; a < p.n_alerts; ++a) {
            const unsigned int* w = reinterpret_cast<const unsigned int*>(p.alerts + (long long)a * kRecBytes);
            const int dst = (int)w[3];
            const unsigned int w4 = w[4];
            const int before = total;
            if ((unsigned)dst < (unsigned)p.n_nodes)
                exact_apply(d, s, &dummy_pend, dst, w4 & d.kmask, ((w4 >> 16) & 0xFFu) != 0, lane, p.out_idx, p.out_cap,
                            &total);
            s.npend = 0;
            if (lane == 0) p.out_counts[a] = total - before;
        }
    } else {
        exact_batch_end(d, s, &dummy_pend, lane, p.out_idx, p.out_cap, &total, &n_applied, &n_full);
        if (lane == 0) p.out_counts[0] = total;
    }
    __syncthreads();
    if (lane == 0) {
        p.scal[0] = s.running;
        p.scal[1] = s.proposal_count;
        p.scal[2] = s.seen_down ? 1 : 0;
        *p.out_n = total;
    }
}

}  // namespace rapid
