// Alert-tally kernels for gfx950 (CDNA4, wave64) -- the hot path.
//
// What they compute (reference semantics, R/ = /root/reference/rapid/src/main/java/com/vrg/rapid/):
//   per receiver: MembershipService.handleMessage(BatchedAlertMessage)  R/MembershipService.java:300-354
//                 + filterAlertMessages                                 R/MembershipService.java:644-675
//                 + MultiNodeCutDetector.aggregateForProposal           R/MultiNodeCutDetector.java:76-128
//                 + MultiNodeCutDetector.invalidateFailingEdges         R/MultiNodeCutDetector.java:137-164
//
// Mapping onto the machine:
//   * one wavefront per simulated receiver, persistent; a workgroup is W such wavefronts (one workgroup per CU, W
//     chosen by the host) that share read-only per-round tables in LDS; receivers are dealt to WORKGROUPS statically
//     (b, b + G, ...) and claimed by the workgroup's waves from a counter in LDS;
//   * the receiver's whole detector state is one word per SLOT in LDS -- a slot is a "hot" subject: one the round's
//     alert set names on >= L distinct rings, the only kind that can ever reach the L watermark at any receiver
//     (index_kernels.h builds the node->slot dictionary once per round; reports about other subjects can never change any
//     receiver's outcome and only contribute to seenLinkDownEvents): bits 0..K-1 = rings reported, bit 16 = already
//     flushed into an emitted proposal;
//   * the delivered stream is read ONCE from HBM, straight into registers (stream_load.h), by this kernel and by nothing else
//     (round 3 moved 68 bytes per record through a load / split pass, a resolve pass and the tally; since round 4 the record
//     stays what it is when it crosses the C ABI).  kFmtBoundary -- the product path of loaded and attached streams -- reads the
//     20-byte rapid_alert_record itself: lane l of quarter q loads {configuration id} and {dst, ring mask | status | flags} of
//     record 64 q + l with two buffer_load_dwordx2 at a lane stride of 20 B (src shares their cache lines and never reaches a
//     register: R/MultiNodeCutDetector.java:101 never reads it), and the record becomes {subject, core word} on its way
//     through the registers (open()): the configuration-id compare of R/MembershipService.java:653-657, the status byte as
//     two bits, the batch end in bit 16; the subject is then mapped to its DICTIONARY ENTRY (dict_entry below: slot, rings the
//     index was not built for, which edge status fails the membership filter) by a lookup whose tables live in LDS (direct or
//     compressed), in memory, or -- hashed -- in LDS again (kDictMode).  The core word (core_word below) is laid out against
//     the entry so that "this delivery is not covered" is one AND and "apply it" is one ds_or of the whole word.
//     kFmtResident -- what rapid_sim_generate writes when it resolves while it lays down -- is 8 bytes {entry, core word}
//     and needs no lookup (kDictResolved).  A WINDOW is kQ quarters of 64 records at fixed positions of the stream; two
//     windows of boundary records per wave are in flight (three of resident ones, four in the packed instantiations), each
//     register set re-requested in place as soon as its window is applied, without a byte of LDS;
//   * FAST window (the steady state): while a WITNESS exists -- a slot in preProposal that provably stays below H
//     through the window even if it is credited with every implicit report it can ever get -- updatesInProgress
//     stays >= 1, so the reference cannot emit inside the window (R/MultiNodeCutDetector.java:110-121) and the
//     window's reports are applied order-free with ds_or_b32, nothing returned, nothing counted.  The witness is
//     checked BEFORE anything is applied (its state, kept in a scalar, + the window's own reports about it), so nothing
//     is ever rolled back.  The records after the window's last batch end are CARRIED (one register) into the next
//     window, so the applied state always stands at a batch boundary.  Implicit reports are owed and applied by one
//     pass over the round's (subject, observer, ring) triples among the hot slots when the fast path is left.  Fast (and
//     cold) windows run in a loop of their own, three per turn, that holds nothing but what such a window needs;
//   * COLD window (before any subject has reached L): applied with ds_or_rtn and kept iff no subject it touches can
//     reach H even with every implicit report it can ever get; its entrants are the first witnesses;
//   * SLOW window (no witness survives, the stream's last window, test modes): the window is decoded into an LDS
//     scratch list and taken by the CAREFUL path (<= 64 records at a time, invalidation applied immediately, exact
//     crossing count, halving) and finally by the EXACT path, record by record;
//   * after the batch that announces a proposal the receiver ignores the rest of its stream
//     (announcedProposal, R/MembershipService.java:318-319) -- the wave stops reading it.
//
// No MFMA: the path is integer scatter/popcount.  All LDS is carved from the 16-byte aligned dynamic segment.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#ifdef RAPID_TEST_BUILD
#include <lds_dma.h>  // (what the LDS-DMA stream probe of stream_probes.inc is written with; test build only)
#endif
#include <stream_load.h>

namespace rapid {

constexpr int kWave = 64;
constexpr int kRecBytes = 20;   // a rapid_alert_record as it crosses the boundary
constexpr long long kMaxStreamRecords = ((1ll << 32) - (1ll << 20)) / 20;  // one receiver's stream is addressed with 32-bit byte offsets
constexpr int kCoreBytes = 8;   // what a tally launch reads per delivered record: {the subject's dictionary entry (or, in the cross-check
                                // modes, its node index), core word} -- beside them, read by no launch: the subject again and the
                                // configuration id (src is never read: R/MultiNodeCutDetector.java:101)
#ifndef RAPID_QUARTERS
#define RAPID_QUARTERS 4
#endif
constexpr int kQ = RAPID_QUARTERS;        // quarters (64 records each) per window
constexpr int kWin = kQ * kWave;          // 256 records = 5 KiB of stream per window
static_assert(kQ >= 1 && kQ <= 16, "window size");
// The PACKED instantiations (rounds with thousands of hot subjects: one wave per SIMD, its LDS) take windows twice as large: what a
// window costs beside its records -- the certificate, the scalar bookkeeping, a branch or two -- is latency that one wave per SIMD
// cannot hide, and it is paid per window (C5: 1.28 -> 1.15 ms per 152 M boundary records with windows of 512 and three sets in
// flight, profiles/r06_c5_window_quarters.txt).  (A build with another RAPID_QUARTERS -- the emulator's q1 / q2 / q3 variants --
// uses that for both.)
#ifndef RAPID_QUARTERS_PACKED
#define RAPID_QUARTERS_PACKED (RAPID_QUARTERS == 4 ? 8 : RAPID_QUARTERS)
#endif
constexpr int kQPacked = RAPID_QUARTERS_PACKED;
static_assert(kQPacked >= 1 && kQPacked <= 16, "window size (packed)");
// ... the packed instantiations over BOUNDARY records, that is.  The packed tally of resolved records runs beside the generator that
// makes them (rapid_sim_round_tiled), and there a SIMD holds four generator waves of 88 registers next to a tally wave of at most
// 160: with windows of 512 records that tally is faster by a seventh alone (180 registers) and the shard round slower by a
// seventh (99.5 -> 113.4 ms: the generator, one wave short, is the critical path).  It keeps windows of 256.
__host__ __device__ constexpr int tally_quarters(bool packed, bool boundary) { return packed && boundary ? kQPacked : kQ; }
// (one LDS layout for both packed forms: the scratch list is sized for the larger window)
__host__ __device__ constexpr int tally_scratch_words(bool packed) { return ((packed && kQPacked > kQ ? kQPacked : kQ) + 1) * kWave; }  // carried quarter + window, one decoded word per record
#ifndef RAPID_UNDO_CAP
#define RAPID_UNDO_CAP 128
#endif
constexpr int kUndoCap = RAPID_UNDO_CAP;         // implicit bits set inside one careful sub-chunk (more: the receiver is redone on the exact path)
constexpr int kCand = 4;                         // witness candidates kept from one sweep
constexpr int kClaimAhead = 3;                   // windows before the end of a stream at which the next receiver is claimed
#ifndef RAPID_DUMMY_SLOTS
#define RAPID_DUMMY_SLOTS 64
#endif
constexpr int kDummySlots = RAPID_DUMMY_SLOTS;   // slots n_hot .. n_hot + 63: where reports about subjects that are not hot go
static_assert((kDummySlots & (kDummySlots - 1)) == 0 && kDummySlots >= 1 && kDummySlots <= 64, "dummy slots");
constexpr int kMaxWavesPerBlock = 16;

// The core word of a delivered record: dword 4 of the boundary record {ring mask, status, flags} converted on the record's way
// through the registers (open() / core_word()), or stored as the second dword of a generated resident record: the status as TWO
// bits, exactly one of which is set in a real record and none in the zeros behind a stream's end, so that "this report fails the
// UP / DOWN filter for this subject" is one AND with the subject's dictionary entry; the batch end in bit 16 -- the low bit
// of the word's upper half, which nothing else of the word reaches into: a lane counts the batch ends it has applied by
// adding upper halves (one vector instruction per quarter; counting them in scalars took two more per quarter out of the one
// scalar pipe the CU's waves share, the busiest unit of this kernel).
constexpr unsigned int kCoreRings = 0x3FFFu;   // bits 0..13: ring mask (K <= 14)
constexpr unsigned int kCoreDown = 1u << 14;   // edgeStatus == DOWN
constexpr unsigned int kCoreUp = 1u << 15;     // edgeStatus == UP
constexpr unsigned int kCoreEob = 1u << 16;    // last record of its BatchedAlertMessage
// (bit 31 of a subject: what the CPU emulator's harness marks a record of another configuration with when it prepares resident
// records by hand -- out of every node range, so that the record takes the path of any report about an unknown node.  The
// kernels compare the configuration id of a boundary record themselves and never see the mark.)
constexpr unsigned int kCoreStale = 1u << 31;
__host__ __device__ inline unsigned int core_word(unsigned int boundary_dword4) {
    return (boundary_dword4 & kCoreRings) | ((boundary_dword4 & 0x00FF0000u) != 0u ? kCoreDown : kCoreUp) |
           ((boundary_dword4 & 0x01000000u) != 0u ? kCoreEob : 0u);
}

// dictionary entry (16 bit): bit 15 = node is a member, bit 14 = slot has hot adjacency, bits 0..13 = slot
constexpr unsigned int kDictMember = 1u << 15;
constexpr unsigned int kDictHasAdj = 1u << 14;
constexpr unsigned int kSlotMask = 0x3FFFu;
constexpr unsigned int kNoSlot = 0x3FFFu;            // at most 16318 hot subjects per round (+ 64 dummy slots)

// What the tally looks up per record -- ONE 32-bit word per node (staged in LDS in the direct mode, assembled from the
// index's tables in the others), laid out against the core word so that (core & entry) & 0xFFFF != 0 <=> the report is
// not covered by what the round index was built for:
//   bits 0..13  rings the round's alert set does NOT name for the node (none for a hot one),
//   bit 14      a DOWN report about the node fails the filter of R/MembershipService.java:659-668 (it is not a member),
//   bit 15      an UP report fails it (it is a member),
//   bits 16..30 2 x slot (a dummy slot for a node that is not hot): shifted right by 16 and added twice it is the byte
//               offset of the slot's state word; bit 16 is therefore 0 (the batch-end bit of the core word meets nothing).
__host__ __device__ inline unsigned int dict_entry(unsigned int decl_entry, unsigned int slot) {
    return (~decl_entry & kCoreRings) | ((decl_entry >> 15) != 0u ? kCoreUp : kCoreDown) | (slot << 17);
}
constexpr unsigned int kEntryPoison = kCoreRings | kCoreDown | kCoreUp;  // no report about such a node is covered (| slot << 17)

// decoded record (scratch list, carry): bits 0..13 ring bits to apply (0: fails the filter or subject not hot),
// bits 14..27 slot, bit 28 = a DOWN report that passed the filter, bit 29 = last record of its batch
constexpr unsigned int kDecDown = 1u << 28;
constexpr unsigned int kDecEob = 1u << 29;

// Per-round index over the loaded alert set (built by index_kernels.h; all device pointers).
// Slots [0, n_hot) are the "hot" subjects -- those named on >= L distinct rings by the round's alert set, the only
// ones that can ever enter preProposal/proposal at any receiver -- in ascending node order.
struct RoundIndex {
    const unsigned short* dict;     // [n_nodes] node -> kDictMember | kDictHasAdj | slot (kNoSlot: not hot)
    const unsigned short* decl;     // [n_nodes] rings the round's alert set names for the node (all rings for a hot one) | member << 15
    // the same two tables in compressed form, for populations whose direct tables do not fit the LDS: a node is TOUCHED if
    // the round's alert set names it at all; tbits = one bit per node, trank[w] = touched nodes before word w,
    // tent[rank] = dict_entry(decl entry, slot) of the rank-th touched node (slot kNoSlot: touched but not hot)
    const unsigned int* tbits;      // [(n_nodes + 31) / 32]
    const unsigned short* trank;    // [(n_nodes + 31) / 32]
    const unsigned int* tent;       // [n_touched]
    int n_touched;
    // dict_entry per node, [n_nodes] = the entry out-of-range subjects are sent to (written once per round by the index:
    // index_fused_kernel, or dict_entries_kernel behind the other forms; 16-byte aligned).  Direct mode: copied into LDS at the
    // head of the launch.  Tables in memory: ONE gather per record (the entries of the round's hot subjects, all that most
    // records ever ask for, stay in the caches).
    const unsigned int* entries;
    // kDictHashed (rounds with thousands of hot subjects over populations of up to 2^21 nodes, every named subject hot): the hot
    // subjects as an EXACT dictionary in ~1.7 bytes per key, small enough for the LDS next to the receivers' state.  A node's key is
    // x = (node * hmul) mod 2^hbits (hmul odd: a bijection of the hbits-bit numbers; hbits covers n_nodes), its bucket x >> 8, its
    // remainder x & 255; the slots are numbered by (bucket, remainder), hoff[b] = first slot of bucket b (hoff[buckets] = n_hot),
    // hrem[slot] = the slot's remainder (one byte; 32 bytes of padding behind the last), hmem bit slot = the node is a member.
    // A bucket holds at most kHashBucketCap keys (index_hash_kernel tries multipliers until that holds).  (bucket, remainder)
    // determine the node, so a match is the node itself, never a look-alike.
    const unsigned short* hoff;     // [hbuckets + 1]
    const unsigned char* hrem;      // [n_hot + 32]
    const unsigned int* hmem;       // [(n_hot + 31) / 32]
    unsigned int hmul;
    int hbits;
    const int* node_of_slot;        // [n_hot]
    // the hot adjacency: pairs[a] = subject slot | observer slot << 14 | ring << 28 for every (subject, ring, observer) triple
    // among hot slots -- one potential implicit report each (R/MultiNodeCutDetector.java:137-164); smask[slot] = the rings on
    // which a hot observer watches the slot = the implicit reports it can ever receive
    const unsigned int* pairs;      // [n_adj]
    const unsigned short* smask;    // [n_hot]
    int n_hot, n_adj;
};

struct TallyParams {
    // The delivered streams, receiver after receiver: kFmtBoundary -- the 20-byte rapid_alert_records exactly as they crossed the C
    // ABI (loaded, attached in place, or generated), 20 bytes pulled from HBM per delivered record, once; kFmtResident -- the
    // 8-byte {dict_entry of the subject, core word} records of rapid_sim_generate(RAPID_GEN_RESOLVED).  (cfg: unused by the
    // kernels -- the emulator's harness keeps the configuration ids of hand-made resident records there.)
    const unsigned char* core;         // [n_records][20 or 8]
    const unsigned char* cfg;
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
    unsigned int* error_flags;        // sticky: bit0 = a delivered report is not covered by the index (see RoundIndex::decl),
                                      // bit1 = a receiver's stream offsets do not lie inside the records (stream_bytes)
    // Readable bytes at `core`.  The offsets of an attached stream set are the caller's device array and are checked HERE, by the
    // wave that is about to follow them -- nothing is copied to the host and no kernel runs ahead of this one: a stream whose
    // offsets are not 0 <= rec_off[r] <= rec_off[r + 1], longer than kMaxStreamRecords, or past stream_bytes is not read at all
    // (the receiver gets "no proposal") and the round's results are void (RAPID_EINVAL when they are asked for).
    unsigned long long stream_bytes;
    int waves_per_block;
    // Receivers [0, n_static) are dealt to the workgroups statically (n_static is a multiple of the grid size); the rest
    // is a common pool claimed through pool[0] by waves whose workgroup has worked off its own deal -- workgroups do not
    // run equally fast (their finishing times spread by ~ +-10 %), and the pool lets the fast ones take the difference.
    // pool[1] counts finished workgroups; the last one zeroes both words for the next launch.  n_static = n_receivers and
    // pool = nullptr: everything dealt statically.
    int n_static;
    unsigned int* pool;
    // Fast-round vote statistics gathered while the proposals are written (R/FastPaxos.java:125-156 counts identical
    // proposals): vote_acc[2] = voters, [3] = max of ~(index of a voter) (= ~lowest voter), [4] = finished workgroups
    // ([0], [1] unused); all zero between launches.  The last workgroup turns them into vote_res[0..9] -- [0] = the lowest
    // voter (0xFFFFFFFF: nobody voted), [2] = voters, the rest zero, in the layout vote_verify_kernel (vote_kernels.h)
    // completes: it takes the lowest voter's proposal as the candidate and counts + verifies the voters that hold it,
    // which settles every round in which that candidate has a quorum without a counting pass -- and zeroes them again.
    // nullptr: not gathered.
    unsigned long long* vote_acc;
    unsigned long long* vote_res;
    // bitmaps[r * bitmap_words + i] = which of the hot slots 64 i .. 64 i + 63 receiver r proposes (written for receivers that
    // announce; bitmap_words = ceil(n_hot / 64)): the proposal in the round's own slot numbering, 64 bytes per receiver at C3b
    // instead of a 2 KB node list -- what the fast-round verification compares element for element (vote_kernels.h).  nullptr:
    // not written.
    unsigned long long* bitmaps;
    int bitmap_words;
    // Every other wave of a workgroup starts `stagger` x 8,128 cycles late.  All receivers of a round cost the same and all
    // waves start together, so without it the waves of a launch move in step: everybody streams (the memory system
    // saturated), then everybody tallies the end of a stream and writes results (the memory system idle).
    int stagger;
    int flags;                        // bit0: exact path only, bit3: careful path only (both for tests); bit5: stream only
    // The fast round SETTLED INSIDE THE LAUNCH (R/FastPaxos.java:125-156 counts identical proposals; a proposal with
    // N - floor((N-1)/4) > N/2 votes is the only one that can have a quorum).  Every workgroup notes which of its receivers voted; at
    // its end -- its waves have stopped streaming -- it looks for the round's CANDIDATE: the first workgroup to get there claims
    // vote_cand[0] by compare-and-swap for its first voter and publishes that proposal's fingerprint, size and bitmap; every other
    // workgroup compares its voters' bitmaps with the published one, word for word -- nothing is counted on fingerprints alone.  A
    // workgroup that arrives inside the microsecond between claim and publication hands its voters to vote_deferred, and the launch's
    // LAST workgroup compares those, then lays the answer down in vote_res[0..9] + the candidate's node list behind it -- exactly the
    // block vote_verify_kernel (vote_kernels.h) used to complete in a launch of its own (12 us on the round's path at C3b) -- and, for
    // a population held by one rank, into the host-mapped page the host polls (vote_publish / vote_seq_out).  Which voter's proposal
    // is the candidate changes no result: a candidate with a quorum is decided; a candidate without one among voters that do not all
    // hold it sends the round to the exact plurality count, as before.
    //   vote_cand: [0] = 1 + the candidate's receiver (0: nobody has voted yet), [1] = published, [2] = fingerprint, [3] = its
    //   prop_count, [4 .. 4 + bitmap_words) = its bitmap; all zero between launches.  vote_acc[0] = voters holding the candidate's
    //   fingerprint whose bitmap differs, vote_acc[1] = voters holding it.  nullptr (or bitmap_words > 64): not settled here.
    unsigned long long* vote_cand;
    unsigned int* vote_deferred;      // [0] = how many, [1 .. 1 + vote_deferred_cap) = receivers; [0] zero between launches
    int vote_deferred_cap;
    volatile unsigned long long* vote_publish;
    volatile unsigned int* vote_seq_out;
    unsigned int vote_seq;
};

__host__ __device__ inline int align16(int x) { return (x + 15) & ~15; }
// LDS budget.  Shared: the node -> slot dictionary and the declared ring masks (direct: 4 B per node; compressed: 3 bits per
// node + 4 B per touched node; or left in memory), the round's (subject, observer, ring) triples [count, triples ...], the per-slot masks of the rings on
// which a hot observer watches the slot, and slot -> node.  Per wave: detector state (hot + dummy slots),
// decoded-record scratch, undo list.
// dictionary placement: where node -> (slot, declared rings, member) is looked up, one instantiation each
//   kDictDirect     one dict_entry per node in LDS, staged from the index's finished entries[] (the product path up to ~30,000 nodes)
//   kDictCompressed one bit per node + rank + one entry per node the alert set names, in LDS (10^5 nodes in 25 KB)
//   kDictMemory     entries[] gathered through L2 (rounds whose state leaves the LDS no room: the packed detector, 10^6 nodes)
//   kDictHashed     the hot subjects as hashed buckets of one-byte remainders in LDS, exact (packed rounds; opt-in, see engine.hip)
//   kDictResolved   no lookup: a generated resident record carries its subject's entry (rapid_sim_generate resolves each of the
//                   round's distinct alerts once, while the deliveries are laid down)
enum { kDictMemory = 0, kDictDirect = 1, kDictCompressed = 2, kDictResolved = 3, kDictHashed = 4 };
// the hashed dictionary's geometry: key bits (>= 10, covering n_nodes), one-byte remainders, at most kHashBucketCap keys per bucket
constexpr int kHashRemBits = 8, kHashBucketCap = 16, kHashMaxKeyBits = 21, kHashPad = 32;
__host__ __device__ inline int hash_key_bits(int n_nodes) {
    int b = 10;
    while (b < 31 && (1ll << b) < (long long)n_nodes) ++b;
    return b;
}
__host__ __device__ inline int hash_buckets(int n_nodes) { return 1 << (hash_key_bits(n_nodes) - kHashRemBits); }
__host__ __device__ inline int tally_dict_bytes(int mode, int n_nodes, int n_touched) {
    if (mode == kDictDirect) return align16((n_nodes + 1) * 4);  // one dict_entry per node + the entry out-of-range subjects are sent to
    if (mode == kDictCompressed) return align16(((n_nodes + 31) / 32) * 4) + align16(((n_nodes + 31) / 32) * 2) + align16(n_touched * 4);
    if (mode == kDictHashed)  // (every named subject is hot in such a round: n_touched == n_hot)
        return align16((hash_buckets(n_nodes) + 1) * 2) + align16(n_touched + kHashPad) + align16(((n_touched + 31) / 32) * 4);
    return 0;
}
// slot -> node (read once per proposed node, when a receiver's proposal is written) stays in memory when a round has so
// many hot subjects that the LDS is better spent on receivers: C5's ~15,000 hot subjects per round at N = 10^6
// ... and so do the per-slot masks of the rings on which a hot observer watches the slot (read by cold windows, sweeps and the
// choice of a witness, never by a fast window).  RAPID_SLOT_TABLES_LDS_MAX hot subjects is where a round goes over to the packed
// detector state (tally_wants_packed), and the packed instantiations are the ones that read the per-slot tables from memory.
#ifndef RAPID_SLOT_TABLES_LDS_MAX
#define RAPID_SLOT_TABLES_LDS_MAX 4096
#endif
constexpr int kSlotNodesInLdsMax = RAPID_SLOT_TABLES_LDS_MAX;
// (the per-slot tables are in LDS exactly when the detector state is not packed: the property of the INSTANTIATION, so that no
// access to them has to choose between an LDS and a memory pointer at run time -- see l_smask in the kernel)
__host__ __device__ inline int tally_shared_bytes(int mode, int n_nodes, int n_touched, int n_hot, int n_adj, bool packed = false) {
    return tally_dict_bytes(mode, n_nodes, n_touched) + align16((n_adj + 1) * 4) +
           (!packed ? align16((n_hot + kDummySlots) * 2) + align16(n_hot * 4) : 0);
}
// per-workgroup statistics accumulator at the very end of the dynamic LDS segment
// eight counters, the workgroup's claim counter, four vote accumulators (112 bytes); then, for the fast round settled inside the
// launch (TallyParams::vote_cand): four flag words and the receivers of this workgroup that voted
constexpr int kBlockVoters = 160;
constexpr int kBlockStatsBytes = 112 + 16 + kBlockVoters * 4;
// packed: two slots per LDS word (PackedSlotDetector) -- rounds with thousands of hot subjects, where the detector state decides
// how many receivers a CU holds (C5: 15,000 hot subjects = 60 KB per receiver as 32-bit words)
__host__ __device__ inline int tally_state_bytes(int n_slots, bool packed) { return align16((n_slots + kDummySlots) * (packed ? 2 : 4)); }
__host__ __device__ inline int tally_wave_bytes(int n_slots, bool packed = false) {
    return tally_state_bytes(n_slots, packed) + tally_scratch_words(packed) * 4 + kUndoCap * 4;
}
// rounds with more hot subjects than the per-slot tables' LDS limit run packed (and with their dictionary in memory)
__host__ __device__ inline bool tally_wants_packed(int n_hot) { return n_hot > kSlotNodesInLdsMax; }

// the workgroup's dynamic LDS segment (every `extern __shared__` array of a kernel is the same memory)
__device__ __forceinline__ unsigned char* dynamic_lds() {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    return smem;
}

// A pointer into the workgroup's LDS that carries its address space in its type (see the per-slot tables of the tally kernel).
// tests/emu/ compiles these headers with g++, where there is one address space: a plain pointer.
#if defined(__clang__) && defined(__HIP__)
template <class T>
using lds_ptr = __attribute__((address_space(3))) T*;
template <class T>
__device__ __forceinline__ lds_ptr<T> to_lds(void* generic) {  // (the low half of a generic LDS address is the offset in the segment)
    return (lds_ptr<T>)(unsigned long long)generic;
}
#else
template <class T>
using lds_ptr = T*;
template <class T>
inline lds_ptr<T> to_lds(void* generic) {
    return static_cast<T*>(generic);
}
#endif

// ---- small wave helpers ---------------------------------------------------------------------------------------
// A receiver is owned by ONE wavefront; that wave's LDS operations execute in program order, so cross-lane
// hand-offs through its private LDS region only need the compiler not to reorder or cache them -- no s_barrier.
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ unsigned long long lanes_lt(int lane) { return (1ull << lane) - 1ull; }
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
__device__ __forceinline__ unsigned int wave_min32(unsigned int v) {
    for (int off = 32; off > 0; off >>= 1) v = min(v, (unsigned int)__shfl_xor((int)v, off, kWave));
    return uniform(v);
}
__device__ __forceinline__ unsigned int wave_or32(unsigned int v) {
    for (int off = 32; off > 0; off >>= 1) v |= (unsigned int)__shfl_xor((int)v, off, kWave);
    return uniform(v);
}
__device__ __forceinline__ unsigned long long mix64(unsigned long long x) {  // splitmix64 finaliser
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// Measurement builds hook into the kernels below through these macros and nothing else.  Their bodies -- phase timers, workgroup
// time stamps, the emulator's trace lines -- live in tally_probes.inc, which only a build that says -DRAPID_MEASUREMENT_BUILD ever
// sees (scripts/build_variants.sh; engine.hip refuses to compile the PRODUCT with it, tests/test_build.py checks both).  Here: the
// empty defaults -- what librapid_mi355x.so and its test build are compiled with.  (The timing-only probes of rounds 4 and 5 --
// the turn loop without its lookup, its OR, its configuration ids -- are gone from the source: what they measured is in
// profiles/r04_ab_probes_c3b.txt, r04_ab_sensitivity_c3b.txt, r05_c5_probe_builds.txt, r05_ab_dummy_slots_c3b.txt.)
#ifdef RAPID_MEASUREMENT_BUILD
#include "tally_probes.inc"
#else
#define RAPID_T0(v)                      // start of a timed phase
#define RAPID_T1(acc, v)                 // end of it
#define RAPID_HOOK_BLOCK_INIT()          // next to the zeroing of the workgroup's statistics
#define RAPID_HOOK_KERNEL_LOCALS()       // a wave's own accumulators
#define RAPID_HOOK_RECEIVER_BEGIN()      // first statement of a receiver
#define RAPID_HOOK_TIGHT(n)              // n windows went through the steady-state loop
#define RAPID_HOOK_RESULTS()             // lane 0, after the receiver's results were stored
#define RAPID_HOOK_RECEIVER_END()        // last statement of a receiver
#define RAPID_HOOK_STATS()               // after mine_stats[] was filled
#define RAPID_HOOK_REDUCE(i) false       // true: the hook folded mine_stats[i] into block_stats[i] itself
#define RAPID_HOOK_STATS_OUT(i) false    // true: the hook wrote block_stats[i] to p.stats itself
#define RAPID_TRACE_LINE(...)            // a line of the emulator's trace
#endif

// ---- detector state accessors ---------------------------------------------------------------------------------
// LDS flavour (population kernel): indices are slots; sweeps cover the hot slots only.
struct SlotDetector {
    // one 32-bit word per slot: bits 0..K-1 = rings reported, bit 17 = already flushed into an emitted proposal.  The fast
    // window ORs whole core words into it, so bits 14, 15 and 16 (status, batch end) hold garbage: every reader masks.
    static constexpr unsigned int kFlushed = 1u << 17;
    unsigned int* st;
    int n_scan;        // = n_hot
    int H, L;
    unsigned int kmask;
    __device__ __forceinline__ unsigned int load(int i) const { return st[i]; }
    __device__ __forceinline__ void store(int i, unsigned int v) const { st[i] = v; }
    __device__ __forceinline__ int count(unsigned int m) const { return __popc(m & kmask); }
    __device__ __forceinline__ unsigned int or_bits(int i, unsigned int bits) const { return atomicOr(&st[i], bits); }
    __device__ __forceinline__ void clear_bits(int i, unsigned int bits) const { atomicAnd(&st[i], ~bits); }
    __device__ __forceinline__ void sync() const { wave_lds_fence(); }
    // the fast window's application: the WHOLE core word into the slot whose doubled number is `so` (= entry >> 16), nothing returned
    __device__ __forceinline__ void fast_or(unsigned int so, unsigned int core) const {
        (void)atomicOr(reinterpret_cast<unsigned int*>(reinterpret_cast<unsigned char*>(st) + 2u * so), core);
    }
    __device__ __forceinline__ void attach(unsigned char* base) { st = reinterpret_cast<unsigned int*>(base); }
};

// Two slots per LDS word: bits 0..13 = rings reported, bit 14 = already flushed into an emitted proposal, in each half.  Reads
// and plain stores are 16-bit LDS accesses; the order-free ORs go to the containing word with the ring bits shifted into the
// slot's half -- three more vector instructions per quarter of a fast window than the 32-bit layout, for half the LDS.
struct PackedSlotDetector {
    static constexpr unsigned int kFlushed = 1u << 14;
    unsigned short* st16;
    unsigned int* st32;
    int n_scan;
    int H, L;
    unsigned int kmask;
    __device__ __forceinline__ unsigned int load(int i) const { return st16[i]; }
    __device__ __forceinline__ void store(int i, unsigned int v) const { st16[i] = (unsigned short)v; }
    __device__ __forceinline__ int count(unsigned int m) const { return __popc(m & kmask); }
    __device__ __forceinline__ unsigned int or_bits(int i, unsigned int bits) const {
        const int sh = (i & 1) * 16;
        return (atomicOr(&st32[i >> 1], bits << sh) >> sh) & 0xFFFFu;
    }
    __device__ __forceinline__ void clear_bits(int i, unsigned int bits) const { atomicAnd(&st32[i >> 1], ~(bits << ((i & 1) * 16))); }
    __device__ __forceinline__ void sync() const { wave_lds_fence(); }
    __device__ __forceinline__ void fast_or(unsigned int so, unsigned int core) const {  // so = 2 x slot: the word at byte (so & ~3), the half (so & 2)
        (void)atomicOr(reinterpret_cast<unsigned int*>(reinterpret_cast<unsigned char*>(st32) + (so & ~3u)), (core & kCoreRings) << ((so & 2u) << 3));
    }
    __device__ __forceinline__ void attach(unsigned char* base) {
        st16 = reinterpret_cast<unsigned short*>(base);
        st32 = reinterpret_cast<unsigned int*>(base);
    }
};

// Global-memory flavour (one MultiNodeCutDetector instance, rapid_cd_*): indices are node indices, the implicit
// invalidation walks the view's observer table.  Accesses bypass the per-CU L1 (agent scope) because the atomics
// execute in L2; ordering points drain the vector memory queue.
struct TableDetector {
    static constexpr unsigned int kFlushed = 1u << 14;  // 16-bit state words: ring bits 0..K-1 (K <= 14), bit 14 = flushed
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
    int batch;           // batches fully processed
    int proposal_count;  // getNumProposals()
    bool seen_down;      // seenLinkDownEvents
    bool entered;        // some subject crossed L since the last invalidation pass (a pair may have become applicable)
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
        const bool take = i < d.n_scan && d.count(m) >= d.H && !(m & D::kFlushed);
        if (take) d.store(i, m | D::kFlushed);
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
__device__ inline void exact_apply(const D& d, RxScalars& s, int dst, unsigned int bits, bool down, int lane, int* emit_out,
                                   int emit_cap, int* emit_n) {
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
            s.entered = true;
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

// Implicit-edge invalidation (R/MultiNodeCutDetector.java:137-164) as ONE pass over the round's flat list of
// (subject slot, observer slot, ring) triples among the hot slots: pairs[0] = n, pairs[1 + a] = subject | observer << 14
// | ring << 28.  An implicit report (o -> s, ring k) is applicable iff s is in preProposal and o is in
// proposal U preProposal; both require >= L explicit reports, so only hot slots take part and every triple is one
// potential implicit report.  A few hundred triples, 64 per step.  Returns the number of H crossings caused; logs
// every bit actually set when undo != nullptr.
template <class D>
__device__ inline int invalidate_pairs(const D& d, const unsigned int* pairs, unsigned int* undo, int* n_undo, int lane,
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
        const bool apply = on && cs >= d.L && cs < d.H && co >= d.L && !(mo & D::kFlushed) && !(ms & (1u << k));
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
            const bool apply = o >= 0 && d.count(mo) >= d.L && !(mo & TableDetector::kFlushed) && !(m & (1u << k));
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
// R/MembershipService.java:330.  A pass can only apply something if a subject crossed L since the previous one.
template <class D>
__device__ inline void exact_batch_end(const D& d, RxScalars& s, const unsigned int* pairs, int lane, int* n_applied) {
    if (!s.seen_down || !s.entered) return;
    const int nH = invalidate_pairs(d, pairs, nullptr, nullptr, lane, n_applied);
    s.entered = false;
    s.running -= nH;
    if (nH > 0 && s.running == 0) {
        s.proposal_count++;
        s.batch_emitted = true;
        flush_sweep(d, lane, nullptr, 0, nullptr);
    }
}

// --------------------------------------------------------------------------------------------------------------
// Whole-population tally.  block = waves_per_block x 64; the workgroup's waves claim receivers from its deal (and the pool).
// kDictMode: where a boundary record's subject is mapped to its dictionary entry (see the list above tally_dict_bytes); the host picks
// the one whose tables fit the LDS next to the receivers' state (engine.hip: build_round_index).
// kTrusted: every declared alert of the current configuration passes the filter of R/MembershipService.java:644-675 under the current
// view (checked by the round index) and the deliveries are vouched for as copies of them; the configuration id of every delivered
// record is still compared here, and a delivery that fails the membership filter or names a ring the index was not built for is
// an ERROR of the stream (sticky flag, RAPID_EINVAL) instead of being dropped per delivery.
// kPacked: two slots per LDS word (rounds with thousands of hot subjects).
// --------------------------------------------------------------------------------------------------------------
// Two record formats reach the kernel (TallyParams::core points at either):
//   kFmtBoundary -- the 20-byte rapid_alert_record exactly as it crosses the C ABI (SURVEY 8d's unit): {configuration id, src, dst,
//       ring mask | status | flags}.  THE product path of loaded streams: a delivered record is read from HBM once, by this
//       kernel, and by nothing else -- the configuration-id check of R/MembershipService.java:653-657, the conversion to the core
//       word and the node -> slot lookup (tables in LDS, or through L2 for populations whose tables do not fit) all happen on
//       the record's way through the registers.  Lane l of quarter q loads {cfg id} and {dst, word} of record 64 q + l with two
//       buffer_load_dwordx2 at a lane stride of 20 B (a quarter is 1,280 contiguous bytes; src shares their cache lines and is
//       never loaded into a register: R/MultiNodeCutDetector.java:101 never reads it);
//   kFmtResident -- 8 bytes {the subject's dict_entry, core word}: what rapid_sim_generate writes when the round's deliveries
//       are made on the device (the subjects are resolved while the records are laid down: kDictResolved, no lookup here).
enum { kFmtResident = 0, kFmtBoundary = 1 };
template <int kFmt, int kQ>
struct WindowT {  // kQ x 64 records in flight, lane l of quarter q = record 64 q + l
    unsigned int w3[kQ], w4[kQ];  // resident: dict_entry (or subject); core word
};
template <int kQ>
struct WindowT<kFmtBoundary, kQ> {
    unsigned int c0[kQ], c1[kQ], w3[kQ], w4[kQ];  // configuration id (low, high); dst; dword 4 of the boundary record
};
enum { kApplied = 0, kWitnessFails = 1, kNotFastable = 2 };  // outcome of a fast-window attempt

#ifndef RAPID_SETS
#define RAPID_SETS 3
#endif
#ifndef RAPID_SETS_BOUNDARY
#define RAPID_SETS_BOUNDARY 2
#endif
#ifndef RAPID_SETS_PACKED
#define RAPID_SETS_PACKED 4
#endif
#ifndef RAPID_SETS_PACKED_BOUNDARY
#define RAPID_SETS_PACKED_BOUNDARY 3  // (of 512 records each: kQPacked)
#endif
#ifndef RAPID_SETS_CURRENT
#define RAPID_SETS_CURRENT 3  // boundary records of which {dst, word} only is loaded (kCurrent): a window is as small as a resident one
#endif
// cache policy of the two loads of a boundary record (0 = default, 2 = nt).  Both touch the same cache lines: with nt the second
// one fetches them from L2 again (measured, scripts/micro/boundary_shapes.hip: 5.4 TB/s against 6.1 TB/s for this shape;
// the kernel: 0.436 -> 0.383 ms on C3b), whereas a resident 8-byte record is loaded once and streams best with nt
#ifndef RAPID_BOUNDARY_AUX_A
#define RAPID_BOUNDARY_AUX_A 0
#endif
#ifndef RAPID_BOUNDARY_AUX_B
#define RAPID_BOUNDARY_AUX_B 0
#endif

// Waves per workgroup an instantiation may be launched with (= its register budget: 16 waves per CU leave 128 VGPRs per wave).
// The per-delivery filter over compressed tables on boundary records needs a few more than that: twelve waves (168 VGPRs)
// instead of spills inside the window loop.
// Packed detector state (rounds with thousands of hot subjects): the LDS holds a handful of receivers per CU whatever the register
// budget, so those instantiations are compiled for eight waves per workgroup -- 256 VGPRs per wave -- and keep more windows of the
// stream in flight per wave instead (RAPID_SETS_PACKED): with one or two waves per SIMD the bytes in flight are what a wave itself
// has requested.
__host__ __device__ constexpr int tally_max_waves(int dict_mode, bool trusted, int fmt, bool packed = false) {
    return packed ? 8 : (fmt == kFmtBoundary && dict_mode == kDictCompressed && !trusted) ? 12 : kMaxWavesPerBlock;
}

// RAPID_WAVES_PER_EU (measurement knob): asks the compiler for a register budget that lets that many waves share a SIMD (5: 96
// VGPRs -- two workgroups of ten waves per CU instead of one of fifteen)
#ifdef RAPID_WAVES_PER_EU
#define RAPID_TALLY_OCCUPANCY __attribute__((amdgpu_waves_per_eu(RAPID_WAVES_PER_EU, RAPID_WAVES_PER_EU)))
#else
#define RAPID_TALLY_OCCUPANCY
#endif
// Two proposal bitmaps compared word for word, both read with agent-scope atomic loads (another workgroup, maybe on another XCD, wrote
// them), eight words of each requested together: the comparison is a chain of round trips, not bandwidth.
__device__ inline bool bitmap_differs(const unsigned long long* a, const unsigned long long* b, int words) {
    bool differs = false;
    for (int w0 = 0; w0 < words; w0 += 8) {
        unsigned long long x[8], y[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            x[j] = w0 + j < words ? __hip_atomic_load(a + w0 + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
            y[j] = w0 + j < words ? __hip_atomic_load(b + w0 + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) differs = differs || x[j] != y[j];
    }
    return differs;
}

// kCurrent (pre-validated boundary records only): every alert the round index saw carries the engine's configuration id, and the
// deliveries are KNOWN to be copies of them -- the library laid them down itself, or the caller vouches for it at level 2 of
// rapid_sim_trust_alert_copies (engine.hip: records_known_current; the plain trusted instantiation compares the ids and drops late
// deliveries) -- so does every delivered record, and its configuration id need not travel from the cache line to the registers at all: ONE buffer_load_dwordx2 per record ({dst, word}, non-temporal: nothing asks for the line
// again) instead of two, no 64-bit compare, eight registers fewer per window in flight.  The bytes that cross the HBM interface
// are the same 20 per record (the lines are the same); what is saved is the second request per line between L2 and the CU
// (scripts/micro/boundary_shapes.hip, shape 2 against shape 0: 6.6 against 6.1 TB/s with nothing else going on).
template <int kDictMode, bool kTrusted, int kFmt = kFmtResident, bool kPacked = false, bool kCurrent = false>
__global__ __launch_bounds__(tally_max_waves(kDictMode, kTrusted, kFmt, kPacked) * 64) RAPID_TALLY_OCCUPANCY void tally_population_kernel(TallyParams p) {
    static_assert(!kCurrent || (kTrusted && kFmt == kFmtBoundary), "only vouched-for boundary records can be known to be current");
    static_assert(!kPacked || kDictMode == kDictMemory || kDictMode == kDictResolved || kDictMode == kDictHashed,
                  "packed detector state: dictionary in memory, hashed in LDS, or none");
    static_assert(kDictMode != kDictHashed || kFmt == kFmtBoundary, "the hashed dictionary maps the subjects of boundary records");
    static_assert(kFmt == kFmtResident || kDictMode != kDictResolved, "a boundary record carries its subject, not an entry");
    constexpr bool kTablesInLds = kDictMode == kDictDirect;
    // the window of THIS instantiation (everything below says kQ / kWin / kScratchWords and means these)
    constexpr int kQ = tally_quarters(kPacked, kFmt == kFmtBoundary), kWin = kQ * kWave, kScratchWords = tally_scratch_words(kPacked);
#ifndef RAPID_LEAN_OPEN
#define RAPID_LEAN_OPEN 1
#endif
    constexpr bool kLeanOpen = RAPID_LEAN_OPEN != 0 && kFmt == kFmtBoundary && kTrusted;  // (see fast_try)
    constexpr int kStride = kFmt == kFmtBoundary ? kRecBytes : kCoreBytes;  // bytes from one record to the next
    constexpr unsigned int kQuarterB = (unsigned int)(kWave * kStride);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = (int)(threadIdx.x & 63u);
    const int wave = (int)(threadIdx.x >> 6);
    const int n_hot = p.idx.n_hot;

    // ---- shared read-only tables ----
    // Subjects that are not hot get DUMMY slots n_hot .. n_hot + 63 (spread over the banks): a report about them is ORed
    // into a word nobody reads, so the fast window needs neither a "hot?" test nor an execution mask per record.
    const int dict_bytes = tally_dict_bytes(kDictMode, p.n_nodes, p.idx.n_touched);
    const int pairs_bytes = align16((p.idx.n_adj + 1) * 4);
    const int shared_bytes = tally_shared_bytes(kDictMode, p.n_nodes, p.idx.n_touched, n_hot, p.idx.n_adj, kPacked);
    const unsigned int* tbits = p.idx.tbits;
    const unsigned short* trank = p.idx.trank;
    const unsigned int* tent = p.idx.tent;
    if (kDictMode == kDictCompressed) {
        const int n_words = (p.n_nodes + 31) / 32;
        unsigned int* l_bits = reinterpret_cast<unsigned int*>(smem);
        unsigned short* l_rank = reinterpret_cast<unsigned short*>(smem + align16(n_words * 4));
        unsigned int* l_ent = reinterpret_cast<unsigned int*>(smem + align16(n_words * 4) + align16(n_words * 2));
        // (four copies per thread in flight, as for the direct tables below: the three tables of 10^5 nodes arrive in two or three
        // round trips instead of ten)
        auto stage32 = [&](unsigned int* dst, const unsigned int* src, int n) {
            for (int base = 0; base < n; base += 4 * (int)blockDim.x) {
                unsigned int v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i = base + j * (int)blockDim.x + (int)threadIdx.x;
                    v[j] = i < n ? src[i] : 0u;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i = base + j * (int)blockDim.x + (int)threadIdx.x;
                    if (i < n) dst[i] = v[j];
                }
            }
        };
        stage32(l_bits, p.idx.tbits, n_words);
        stage32(reinterpret_cast<unsigned int*>(l_rank), reinterpret_cast<const unsigned int*>(p.idx.trank), (n_words + 1) / 2);  // (two ranks per word)
        stage32(l_ent, p.idx.tent, p.idx.n_touched);
        tbits = l_bits;
        trank = l_rank;
        tent = l_ent;
    }
    // kDictHashed: bucket bounds, remainders and member bits, staged like the compressed tables (dword copies, four in flight)
    const unsigned short* l_hoff = nullptr;
    const unsigned char* l_hrem = nullptr;
    const unsigned int* l_hmem = nullptr;
    const unsigned int hmask = p.idx.hbits >= 32 ? 0xFFFFFFFFu : ((1u << p.idx.hbits) - 1u), hmul = p.idx.hmul;
    if (kDictMode == kDictHashed) {
        const int nb = hash_buckets(p.n_nodes);
        const int w_off = (nb + 1 + 1) / 2, w_rem = (p.idx.n_touched + kHashPad + 3) / 4, w_mem = (p.idx.n_touched + 31) / 32;
        unsigned int* const d_off = reinterpret_cast<unsigned int*>(smem);
        unsigned int* const d_rem = reinterpret_cast<unsigned int*>(smem + align16((nb + 1) * 2));
        unsigned int* const d_mem = reinterpret_cast<unsigned int*>(smem + align16((nb + 1) * 2) + align16(p.idx.n_touched + kHashPad));
        auto stage32h = [&](unsigned int* dst, const unsigned int* src, int n) {
            for (int base = 0; base < n; base += 4 * (int)blockDim.x) {
                unsigned int v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i = base + j * (int)blockDim.x + (int)threadIdx.x;
                    v[j] = i < n ? src[i] : 0u;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i = base + j * (int)blockDim.x + (int)threadIdx.x;
                    if (i < n) dst[i] = v[j];
                }
            }
        };
        stage32h(d_off, reinterpret_cast<const unsigned int*>(p.idx.hoff), w_off);  // (the global tables are allocated in whole dwords)
        stage32h(d_rem, reinterpret_cast<const unsigned int*>(p.idx.hrem), w_rem);
        stage32h(d_mem, p.idx.hmem, w_mem);
        l_hoff = reinterpret_cast<const unsigned short*>(d_off);
        l_hrem = reinterpret_cast<const unsigned char*>(d_rem);
        l_hmem = d_mem;
    }
    const unsigned int* entries = nullptr;  // direct mode: dict_entry per node, [n_nodes] = where out-of-range subjects are sent
    if (kTablesInLds) {
        // The round index leaves the finished entries in memory (RoundIndex::entries); they are copied 16 bytes at a time, four
        // copies per thread in flight: ONE memory round trip for up to 15,360 nodes.  (Assembled here from the index's two 16-bit
        // tables, entry by entry, the staging of 10^4 nodes was eleven dependent round trips at the head of every launch --
        // ~15 us during which no wave of the workgroup streams.)
        unsigned int* const l_ent = reinterpret_cast<unsigned int*>(smem);
        const int n_ent = p.n_nodes + 1, n_quads = n_ent / 4;
        const uint4* const src4 = reinterpret_cast<const uint4*>(p.idx.entries);
        uint4* const dst4 = reinterpret_cast<uint4*>(l_ent);
        for (int base = 0; base < n_quads; base += 4 * (int)blockDim.x) {
            uint4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = base + j * (int)blockDim.x + (int)threadIdx.x;
                v[j] = i < n_quads ? src4[i] : make_uint4(0u, 0u, 0u, 0u);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = base + j * (int)blockDim.x + (int)threadIdx.x;
                if (i < n_quads) dst4[i] = v[j];
            }
        }
        for (int i = 4 * n_quads + (int)threadIdx.x; i < n_ent; i += (int)blockDim.x) l_ent[i] = p.idx.entries[i];
        entries = l_ent;
    }
    // the hot adjacency (flat list of triples, count first) and the per-slot masks, copied from the round index
    unsigned int* const l_pairs = reinterpret_cast<unsigned int*>(smem + dict_bytes);
    constexpr bool slot_tables_in_lds = !kPacked;  // the per-slot masks and slot -> node (tally_shared_bytes)
    // Where the two per-slot tables live is a property of the instantiation (in LDS unless the detector state is packed), and the
    // LDS copies are read through pointers that SAY they point into the LDS (lds_ptr).  Chosen at run time through generic
    // pointers, "in LDS ? l_smask[slot] : p.idx.smask[slot]" is merged by the compiler into ONE load through a selected pointer
    // -- a FLAT load, which returns out of order with everything else and is waited for with s_waitcnt vmcnt(0); kept apart as
    // two loads under a run-time branch, the wait-count pass still waits with vmcnt(0) before the LDS read (its destination MAY be
    // the pending target of the other branch's load).  Either way every sweep, cold window, choice of a witness and every 64
    // slots of a proposal being written waited for the two windows of the stream in flight and for the stores of the previous
    // 64 slots -- a memory latency each, 11 % of a receiver's cycles in the proposal loop alone (C3b,
    // profiles/r04_phase_timers_out_c3b.txt).
    lds_ptr<unsigned short> const l_smask = to_lds<unsigned short>(smem + dict_bytes + pairs_bytes);
    lds_ptr<int> const l_nos = to_lds<int>(smem + dict_bytes + pairs_bytes + align16((n_hot + kDummySlots) * 2));
    if (threadIdx.x == 0) l_pairs[0] = (unsigned int)p.idx.n_adj;
    {
        // ONE loop on purpose: with three separate copy loops here the register allocation of the whole kernel changes
        // (the per-delivery-filter instantiations go from 116-125 VGPRs to 128 + 104-148 B of scratch per lane, whose
        // reloads sit inside the receiver loop: 0.42 -> 0.52 ms on C3b); tests/test_build.py checks the scratch size.
        const int n_slot_tab = slot_tables_in_lds ? n_hot + kDummySlots : 0;
        const int n_stage = p.idx.n_adj > n_slot_tab ? p.idx.n_adj : n_slot_tab;
        for (int i = (int)threadIdx.x; i < n_stage; i += (int)blockDim.x) {
            if (i < p.idx.n_adj) l_pairs[1 + i] = p.idx.pairs[i];
            if (slot_tables_in_lds) {
                if (i < n_hot) {
                    l_smask[i] = p.idx.smask[i];
                    l_nos[i] = p.idx.node_of_slot[i];
                } else if (i < n_hot + kDummySlots) {
                    l_smask[i] = 0;
                }
            }
        }
    }
    const unsigned int* const pairs = l_pairs;
    constexpr bool nos_in_lds = slot_tables_in_lds;
    auto smask_of = [&](unsigned int slot) -> unsigned int {  // rings on which a hot observer watches the slot (a dummy slot: none)
        if (slot_tables_in_lds) return (unsigned int)l_smask[slot];
        return slot < (unsigned int)n_hot ? (unsigned int)p.idx.smask[slot] : 0u;
    };
    auto smask_uniform = [&](int slot) -> unsigned int {  // the same for a wave-uniform slot: through the scalar cache when the table is in memory
        if (slot_tables_in_lds) return uniform((unsigned int)l_smask[slot]);
        const unsigned int w = stream_scalar_load32(reinterpret_cast<const unsigned int*>(p.idx.smask) + (slot >> 1));
        return (w >> ((slot & 1) * 16)) & 0xFFFFu;
    };
    // The launch statistics are summed per workgroup in LDS and stored once per workgroup: thousands of waves adding to
    // the same eight global words at the end of their lives queue up behind each other in one L2 channel -- measured:
    // 0.13 ms of a 0.63 ms kernel, and every stream that crosses that channel waits with them.
    unsigned long long* const block_stats =
        reinterpret_cast<unsigned long long*>(smem + shared_bytes + (int)(blockDim.x >> 6) * tally_wave_bytes(n_hot, kPacked));
    unsigned int* const block_claims = reinterpret_cast<unsigned int*>(block_stats + 8);  // receivers claimed by this workgroup
    if (threadIdx.x < 8u) block_stats[threadIdx.x] = 0ull;
    unsigned long long* const block_votes = block_stats + 10;  // [4], see TallyParams::vote_acc
    if (threadIdx.x >= 16u && threadIdx.x < 20u) block_votes[threadIdx.x - 16u] = 0ull;
    // [0] = what the workgroup found when it looked for the round's candidate (0: nothing yet, 1: published, 2: claimed but not yet
    // published), [1] = this is the launch's last workgroup, [2] = voters among this workgroup's receivers; then their indices
    unsigned int* const cand_flags = reinterpret_cast<unsigned int*>(block_stats + 14);
    unsigned int* const block_voters = reinterpret_cast<unsigned int*>(block_stats + 16);  // [kBlockVoters]
    if (threadIdx.x >= 20u && threadIdx.x < 24u) cand_flags[threadIdx.x - 20u] = 0u;
    RAPID_HOOK_BLOCK_INIT();
    if (threadIdx.x == 8u) *block_claims = blockDim.x >> 6;  // claims 0 .. waves - 1 are the first deal
    __syncthreads();

    // ---- this wave's private LDS ----
    const int state_bytes = tally_state_bytes(n_hot, kPacked);
    unsigned char* const mine = smem + shared_bytes + wave * tally_wave_bytes(n_hot, kPacked);
    unsigned int* const scratch = reinterpret_cast<unsigned int*>(mine + state_bytes);
    unsigned int* const undo = scratch + kScratchWords;

    typedef typename std::conditional<kPacked, PackedSlotDetector, SlotDetector>::type Det;
    Det d;
    d.attach(mine);
    d.n_scan = n_hot;
    d.H = p.H;
    d.L = p.L;
    d.kmask = (1u << p.K) - 1u;

    const unsigned int node_last = (unsigned int)(p.n_nodes > 0 ? p.n_nodes - 1 : 0);
    const unsigned int my_dummy = (unsigned int)(n_hot + (lane & (kDummySlots - 1)));  // tables in memory: this lane's dummy slot
    unsigned long long n_slow = 0, n_fast = 0, n_restart = 0, n_records = 0, n_pipe = 0, n_careful = 0, n_sweeps = 0;
    RAPID_HOOK_KERNEL_LOCALS();
    int n_applied = 0;
    unsigned int sink = 0u;  // stream-only mode: keeps the loads alive

    typedef WindowT<kCurrent ? kFmtResident : kFmt, kQ> Win;  // (kCurrent: {dst, word} is all that is loaded of a boundary record)
    struct Rec {  // a window as the paths below see it, whatever the format it arrived in
        unsigned int w3[kQ], w4[kQ];  // subject (resident: its dict_entry, or the subject | kCoreStale); core word (core_word)
    };
    struct Stream {  // a receiver's delivered records
        const unsigned char* base;
        unsigned int bytes;
    };
    // one window of the stream `st` starting at this lane's byte offset `voff`: kQ wave instructions, nothing waited for
    auto load_window = [&](const Stream& st, unsigned int voff, Win& W) {
        // declared wave-uniform right here (it is: every lane computes it from the wave's receiver index), so that the
        // descriptor is in SGPRs whatever the compiler concluded about the loops it travelled through
        const unsigned long long b = (unsigned long long)st.base;
        const stream_rsrc_t rsrc = stream_make_rsrc(
            reinterpret_cast<const unsigned char*>(((unsigned long long)uniform((unsigned int)(b >> 32)) << 32) | (unsigned long long)uniform((unsigned int)b)),
            uniform(st.bytes));
#pragma unroll
        for (int q = 0; q < kQ; ++q) {
            if constexpr (kFmt == kFmtBoundary) {
                if constexpr (kCurrent) {
                    stream_load2<2>(rsrc, voff, (unsigned int)q * kQuarterB + 12u, W.w3[q], W.w4[q]);
                } else {
                    stream_load2<RAPID_BOUNDARY_AUX_A>(rsrc, voff, (unsigned int)q * kQuarterB, W.c0[q], W.c1[q]);
                    stream_load2<RAPID_BOUNDARY_AUX_B>(rsrc, voff, (unsigned int)q * kQuarterB + 12u, W.w3[q], W.w4[q]);
                }
            } else {
                stream_load2(rsrc, voff, (unsigned int)q * kQuarterB, W.w3[q], W.w4[q]);
            }
        }
    };
    // A window OPENED: subject and core word per record.  Resident records are stored that way.  A boundary record becomes one
    // here, on its way through the registers: the configuration-id comparison of R/MembershipService.java:653-657 (another id:
    // the alert is dropped -- its rings and status are cleared, its batch end stays, a batch ends whether or not its last alert
    // is dropped), the status byte as two bits, the batch-end flag in bit 16 (core_word).  An alert without ring numbers does
    // nothing in the reference (aggregateForProposal(AlertMessage) iterates over them, R/MultiNodeCutDetector.java:76-82) and is
    // cleared the same way -- which also makes the zeros behind a stream's end the empty record they are in the resident format.
    const unsigned long long cfg64 = (unsigned long long)p.cfg_id;
    auto open = [&](const Win& c) -> Rec {
        Rec x;
#pragma unroll
        for (int q = 0; q < kQ; ++q) {
            if constexpr (kFmt == kFmtBoundary) {
                // (one 64-bit compare instead of two exclusive-ors, an or and a 32-bit compare)
                bool current = true;
                if constexpr (!kCurrent) current = (((unsigned long long)c.c1[q] << 32) | (unsigned long long)c.c0[q]) == cfg64;
                const unsigned int raw = c.w4[q];
                const unsigned int rings = raw & kCoreRings;
                const unsigned int live = current ? rings : 0u;
                const unsigned int eobw = (raw >> 8) & kCoreEob;
                const unsigned int full = rings | ((raw & 0x00FF0000u) != 0u ? kCoreDown : kCoreUp) | eobw;
                x.w3[q] = c.w3[q];
                x.w4[q] = live != 0u ? full : eobw;
            } else {
                x.w3[q] = c.w3[q];
                x.w4[q] = c.w4[q];
            }
        }
        return x;
    };
    // (see TallyParams::stream_bytes) -> the stream's length in records; 0 and the error flag for offsets that cannot be followed
    auto checked_length = [&](long long rec0, long long rec1) -> int {
        const bool ok = rec0 >= 0 && rec1 >= rec0 && rec1 - rec0 <= kMaxStreamRecords &&
                        (unsigned long long)rec1 * (unsigned long long)kStride <= p.stream_bytes;
        if (!ok && lane == 0) stream_flag_or(p.error_flags, 2u);
        return ok ? (int)(rec1 - rec0) : 0;
    };
    auto make_stream = [&](long long rec0, long long rec1) -> Stream {
        Stream st;
        st.base = p.core + (unsigned long long)rec0 * kStride;
        st.bytes = (unsigned int)((rec1 - rec0) * kStride);
        return st;
    };

    // ---- per record: the subject's dictionary entry, and the record's EFFECTIVE core word ----
    // lookup: node -> dict_entry (see there).  Direct mode: one LDS read; an out-of-range subject is clamped onto the poison
    // entry behind the table (nothing about it is covered, and it has a dummy slot).
    struct Look {
        unsigned int entry;
        bool untouched;  // compressed mode: the round's alert set never names the node -- its membership is not in the tables
    };
    const unsigned int n_nodes_u = (unsigned int)(p.n_nodes > 0 ? p.n_nodes : 0);
    auto lookup = [&](unsigned int w3) -> Look {
        Look k;
        k.untouched = false;
        if (kTablesInLds) {
            k.entry = entries[min(w3, n_nodes_u)];
            return k;
        }
        if (kDictMode == kDictResolved) {  // the record carries its subject's entry (poison for a stale or unknown subject)
            k.entry = w3;
            return k;
        }
        if (kDictMode == kDictMemory) {
            k.entry = p.idx.entries[min(w3, n_nodes_u)];
            return k;
        }
        if (kDictMode == kDictHashed) {
            // bucket bounds (two 16-bit reads), the bucket's remainders -- sixteen bytes from its first slot on, fetched as five
            // aligned dwords and shifted into place --, a bytewise compare with the key's remainder, the member bit of the slot
            // that matched.  (x - 0x01010101) & ~x & 0x80808080 marks the zero bytes of x; its lowest mark is exact, and the
            // first match is the only one: the remainders of a bucket are distinct.  A match past the bucket's own slots
            // belongs to a later bucket: no match.
            const bool in = w3 < n_nodes_u;
            const unsigned int x = (w3 * hmul) & hmask;
            const unsigned int b = in ? x >> kHashRemBits : 0u, r = x & 255u;
            const unsigned int o0 = l_hoff[b], o1 = l_hoff[b + 1];
            const unsigned int* const wp = reinterpret_cast<const unsigned int*>(l_hrem + (o0 & ~3u));
            const unsigned int d0 = wp[0], d1 = wp[1], d2 = wp[2], d3 = wp[3], d4 = wp[4];
            const unsigned int sh = (o0 & 3u) * 8u;
            const unsigned int pat = r * 0x01010101u;
            auto zero_marks = [&](unsigned int lo, unsigned int hi) -> unsigned int {
                const unsigned int e = (unsigned int)((((unsigned long long)hi << 32) | (unsigned long long)lo) >> sh) ^ pat;
                return (e - 0x01010101u) & ~e & 0x80808080u;
            };
            const unsigned int z0 = zero_marks(d0, d1), z1 = zero_marks(d1, d2), z2 = zero_marks(d2, d3), z3 = zero_marks(d3, d4);
            const unsigned int idx = z0 != 0u   ? (unsigned int)(__ffs((int)z0) - 1) >> 3
                                     : z1 != 0u ? 4u + ((unsigned int)(__ffs((int)z1) - 1) >> 3)
                                     : z2 != 0u ? 8u + ((unsigned int)(__ffs((int)z2) - 1) >> 3)
                                     : z3 != 0u ? 12u + ((unsigned int)(__ffs((int)z3) - 1) >> 3)
                                                : 0xFFFFu;
            const bool hit = in && idx < o1 - o0;
            const unsigned int slot = hit ? o0 + idx : my_dummy;
            const unsigned int mem = hit ? (l_hmem[slot >> 5] >> (slot & 31u)) & 1u : 0u;
            // a hot node: every ring is declared; which status fails the membership filter follows from the member bit
            k.entry = hit ? ((mem != 0u ? kCoreUp : kCoreDown) | (slot << 17)) : in ? (kCoreRings | kCoreDown | (my_dummy << 17)) : (kEntryPoison | (my_dummy << 17));
            k.untouched = in && !hit;
            return k;
        }
        // kDictCompressed: bit test + rank -- two independent LDS reads -- then the entry of a touched node (its dict_entry, written
        // by the index build; a touched node that is not hot carries kNoSlot there and gets this lane's dummy slot).  A node the
        // alert set never names has no entry: a valid report about it is exactly what the coverage check exists for.
        const bool in = w3 <= node_last && p.n_nodes > 0;
        const unsigned int idx = min(w3, node_last);
        const unsigned int word = tbits[idx >> 5], before = (unsigned int)trank[idx >> 5];
        const unsigned int bit = idx & 31u;
        const bool touched = ((word >> bit) & 1u) != 0u;
        unsigned int e = touched ? tent[before + (unsigned int)__popc(word & ((1u << bit) - 1u))] : (kCoreRings | kCoreDown | (kNoSlot << 17));
        if ((e >> 17) == kNoSlot) e = (e & 0x1FFFFu) | (my_dummy << 17);
        k.untouched = in && !touched;
        k.entry = in ? e : (kEntryPoison | (my_dummy << 17));
        return k;
    };
    // effective: filterAlertMessages (R/MembershipService.java:644-675) applied to record (q, lane) of a window, branch-free:
    // the core word itself if the record passes, with its rings and status cleared if it does not (the batch end stays -- a
    // batch ends whether or not its last alert is dropped).  Past the end of a stream the word is zero.
    //   vouched-for copies of validated alerts (kTrusted): nothing is dropped; what a delivered record can still get wrong --
    //   subject out of range, UP / DOWN against the membership (R/MembershipService.java:659-668), rings the index was not
    //   built for -- is ONE AND with the entry, collected in `uncovered` (sticky error, results void);
    //   otherwise: stale or unknown subject (the poison entry), empty ring list, UP / DOWN against the membership per delivery.
    unsigned int uncovered = 0u;  // per lane: what delivered reports name that the index was not built for
    auto effective = [&](const Rec& c, int q, const Look& k) -> unsigned int {
        const unsigned int w = c.w4[q];
        if (kTrusted) {
            uncovered |= w & k.entry;
            return w;
        }
        // (a subject out of range has the poison entry: BOTH status bits, so whatever status the report carries fails below; a
        // record of another configuration arrives with its rings cleared -- open() -- or, resident, with the poison entry)
        const unsigned int bad0 = (w & kCoreRings) == 0u ? 1u : 0u;
        // (the membership of a node the alert set never names is not in the compressed tables: such a report is
        // not tallied, and flagged if it is otherwise valid)
        const unsigned int bad = bad0 | (w & k.entry & (kCoreDown | kCoreUp)) | (k.untouched ? 1u : 0u);
        const unsigned int weff = bad == 0u ? w : (w & kCoreEob);
        uncovered |= k.untouched ? (bad0 == 0u ? (w & kCoreRings) : 0u) : (weff & k.entry & kCoreRings);
        return weff;
    };
    // the same as (slot, ring bits, DOWN) for the paths that work on decoded records
    struct Dec {
        unsigned int slot, bits;
        bool down;
    };
    auto decode_rec = [&](const Rec& c, int q) -> Dec {
        const Look k = lookup(c.w3[q]);
        const unsigned int w = effective(c, q, k);
        Dec r;
        r.slot = k.entry >> 17;
        r.bits = w & d.kmask;
        r.down = (w & kCoreDown) != 0u;
        return r;
    };
    // decoded word of the scratch list / the carry; reports about subjects that are not hot carry no ring bits there
    auto pack_rec = [&](const Dec& r, bool eob) -> unsigned int {
        const bool hot = r.slot < (unsigned int)n_hot;
        return (hot ? r.bits : 0u) | ((hot ? r.slot : 0u) << 14) | (r.down ? kDecDown : 0u) | (eob ? kDecEob : 0u);
    };

    // Receivers are dealt to WORKGROUPS statically (workgroup b of G takes receivers b, b + G, b + 2 G ...: consecutive
    // receivers go to different CUs) and claimed by the workgroup's waves from a counter in LDS: all receivers of a
    // round cost the same, but the waves do not run equally fast (a wave's share of the memory system varies by ~ +-15 %),
    // and a workgroup ends with its slowest wave.  No word of global memory is touched by every wave of the launch: a
    // global work counter was measured twice (round 1: ~15 us per receiver; round 2, claims spread over time and issued
    // four windows ahead: kernel 0.353 -> 0.386 ms) -- the claims queue up in one L2 channel, and every stream that
    // crosses that channel waits with them.
    const int n_blocks = (int)gridDim.x;
    int r = uniform(wave * n_blocks + (int)blockIdx.x);  // the first deal: claim number `wave` of this workgroup
    if (r >= p.n_static) r = p.n_receivers;              // (the host sizes the static part so that this never takes pool work away)
    // The stream runs kSets windows ahead of the tally: S[0] is the window about to be tallied, S[1 ..] the ones behind it,
    // all requested (a window that is being tallied has kSets - 1 successors in flight -- with one, a wave would wait out a
    // whole memory latency per window as soon as a window's tally is shorter than that, which it is since the fast window
    // shrank to ~100 instructions: 15 waves x 2 KiB per CU are not enough bytes in flight for 8 TB/s).  Three sets of 8 registers.
    // (boundary records: two sets of 16 registers -- 10 KiB of stream in flight per wave against 6 KiB of the resident format)
    constexpr int kSets = kPacked ? (kFmt == kFmtBoundary ? RAPID_SETS_PACKED_BOUNDARY : RAPID_SETS_PACKED) : kCurrent ? RAPID_SETS_CURRENT : kFmt == kFmtBoundary ? RAPID_SETS_BOUNDARY : RAPID_SETS;
    static_assert(kSets >= 2 && kSets <= 6, "window sets");
    constexpr unsigned int kWinBytes = (unsigned int)(kWin * kStride);
    Win S[kSets];
    Stream rsrc;
    rsrc.base = p.core;
    rsrc.bytes = 0u;
    const unsigned int lane_off = (unsigned int)lane * (unsigned int)kStride;  // this lane's byte offset inside a quarter
    if (p.stagger > 0 && (wave & 1) != 0) {  // every other wave starts late (see TallyParams::stagger)
        for (int i = 0; i < p.stagger; ++i) __builtin_amdgcn_s_sleep(127);
    }
    int nrec = 0;
    if (r < p.n_receivers) {
        const long long rec0 = stream_scalar_load(p.rec_off + r), rec1 = stream_scalar_load(p.rec_off + r + 1);
        nrec = checked_length(rec0, rec1);
        rsrc = make_stream(rec0, rec0 + nrec);
#pragma unroll
        for (int i = 0; i < kSets; ++i) load_window(rsrc, lane_off + (unsigned int)i * kWinBytes, S[i]);
    }
    while (r < p.n_receivers) {
        RAPID_HOOK_RECEIVER_BEGIN();
        const int nwin = (nrec + kWin - 1) / kWin;
        // The next receiver is claimed a few windows before the end of this one's stream, and the bounds of its stream are
        // loaded right then: both answers arrive with the stream's last windows instead of costing two memory round trips
        // between two receivers.
        int r_next = p.n_receivers;
        long long next0 = 0, next1 = 0;
        bool claimed = false;
        auto claim = [&]() {
            claimed = true;
            unsigned int v = 0u;
            if (lane == 0) v = atomicAdd(block_claims, 1u);
            r_next = (int)uniform(v) * n_blocks + (int)blockIdx.x;
            if (r_next >= p.n_static) r_next = p.n_receivers;  // the workgroup's own deal is worked off (the pool: after the stream)
            if (r_next < p.n_receivers) {  // scalar loads: no vector register waits for them inside the window loop
                next0 = stream_scalar_load(p.rec_off + r_next);
                next1 = stream_scalar_load(p.rec_off + r_next + 1);
            }
        };

        int emit_batch = -1;
        RxScalars s;
        bool exact_only = (p.flags & 1) != 0;
        bool restart = false;
        int careful_cap = kWave;  // records the careful loop takes at once; narrowed while an emission cannot be excluded
        int hint_cap = -1;        // after a failed careful attempt: records of the sub-chunk before its critical H crossing (-1: unknown)

        // ---- lean-path state ----
        bool cold = true;            // no subject has reached L yet (updatesInProgress = 0, nothing proposed)
        bool owed = false;           // windows were applied since the last invalidation pass: the implicit reports they make possible are owed
        bool running_exact = true;   // s.running is the reference's updatesInProgress
        bool need_sweep = false;     // the witness candidates are stale (a slow window was processed since the last sweep)
        bool swept = false;          // the hot slots have been swept for candidates during the current window
        int witness = -1;            // a slot in preProposal whose bound popc(state | wmask) stays below H
        unsigned int witness_so = 0xFFFFFFFFu;  // 2 x witness, what (entry >> 16) of a report about it looks like (none: matches nothing)
        unsigned int wmask = 0u;     // rings on which the witness can receive an implicit report
        unsigned int wstate = 0u;    // the witness's state word, kept in a scalar: between two set_witness() calls only fast windows
                                     // touch the detector state, and they see every report about the witness they apply
        unsigned int candv = 0u;     // lane i: slot of witness candidate i
        int ncand = 0, ci = 0;
        // the records after the last applied batch end (lanes >= carry_start of the previous window's last quarter): 2 x slot and
        // the effective core word (rings + status; zero in the lanes before carry_start)
        unsigned int carry_so = 0u, carry_w = 0u;
        int carry_start = kWave;
        unsigned int vbatch = 0u;    // per lane: batch ends applied by cold / fast windows since s.batch was last brought up to date

        // per-sub-chunk decode results of the slow path (one record per lane)
        int dst = 0, ncons = 0, lastE = -1, spos = 0, send = 0;
        unsigned int bits = 0;
        bool down = false, eob = false;
        unsigned long long mE_all = 0ull;

        auto set_witness = [&]() {
            witness = ci < ncand ? lane_value((int)candv, ci) : -1;
            witness_so = witness >= 0 ? 2u * (unsigned int)witness : 0xFFFFFFFFu;
            wmask = witness >= 0 ? smask_uniform(witness) : 0u;
            wave_lds_fence();
            wstate = witness >= 0 ? uniform(d.load(witness)) : 0u;
        };
        // ---- one pass over the hot slots: updatesInProgress (slots with L <= count < H), and WITNESS candidates for the
        // fast path: slots in [L, H) that stay below H even if they are given every implicit report they can ever get --
        // popc(state | smask) < H, so that a lagging state cannot hide their departure -- smallest bound first.
        auto sweep = [&]() {
            RAPID_TRACE_LINE("R r=%d\n", r);
            RAPID_T0(ts0);
            wave_lds_fence();
            int run = 0;
            unsigned int best = 0xFFFFFFFFu;  // bound << 16 | slot
            // Per-slot masks in memory (packed detector state: thousands of hot subjects).  Asked for slot by slot under the "in preProposal"
            // mask, every step of the sweep waited out a memory round trip of its own: 118 dependent round trips per sweep at 15,000
            // hot subjects, ~95,000 cycles (profiles/r05_c5_phase_before.txt); round 5 requested the masks of four steps together.
            // Round 6: the packed state is swept EIGHT slots per lane and step -- one ds_read_b128 of the state, one 16-byte load of
            // the masks (the masks of kSweepVec steps requested together), the slots in [L, H) counted per lane and summed once at
            // the end instead of a ballot and a scalar count per slot: 30 steps instead of 118 at 15,000 hot subjects, and the
            // scalar pipe left alone (a sweep was 57,000 cycles, five times per receiver: a fifth of a receiver's time at 10^6 nodes,
            // profiles/r06_c5_phase_resolved.txt).  Which slots a lane sees decides the 2nd to 4th candidate, never what is tallied.
            if constexpr (!slot_tables_in_lds) {
                constexpr int kPer = 8, kStepSlots = kPer * kWave, kSweepVec = 5;
                const uint4* const st128 = reinterpret_cast<const uint4*>(mine);
                const uint4* const sm128 = reinterpret_cast<const uint4*>(p.idx.smask);  // (engine.hip sizes the table with 16 bytes to spare)
                unsigned int run_l = 0u;
                const unsigned int Lu = (unsigned int)d.L, HLu = (unsigned int)(d.H - d.L), Hu = (unsigned int)d.H;
                for (int i0 = 0; i0 < n_hot; i0 += kStepSlots * kSweepVec) {
                    uint4 pm[kSweepVec];
#pragma unroll
                    for (int u = 0; u < kSweepVec; ++u) {
                        const int base = i0 + u * kStepSlots + lane * kPer;
                        pm[u] = sm128[min(base, n_hot - 1) >> 3];  // (clamped, not skipped: loads under a branch cost the wait-count pass its count)
                    }
#pragma unroll
                    for (int u = 0; u < kSweepVec; ++u) {
                        if (i0 + u * kStepSlots >= n_hot) continue;  // (wave-uniform)
                        const int base = i0 + u * kStepSlots + lane * kPer;
                        const uint4 sv = base < n_hot ? st128[base >> 3] : make_uint4(0u, 0u, 0u, 0u);
                        const unsigned int sw[4] = {sv.x, sv.y, sv.z, sv.w}, mw[4] = {pm[u].x, pm[u].y, pm[u].z, pm[u].w};
#pragma unroll
                        for (int j = 0; j < kPer; ++j) {
                            const unsigned int m = (sw[j >> 1] >> ((j & 1) * 16)) & 0xFFFFu, am = (mw[j >> 1] >> ((j & 1) * 16)) & 0xFFFFu;
                            const unsigned int c = (unsigned int)d.count(m);
                            const bool pre = base + j < n_hot && c - Lu < HLu;  // L <= c < H
                            run_l += pre ? 1u : 0u;
                            const unsigned int bound = (unsigned int)d.count(m | am);
                            const unsigned int key = (pre && bound < Hu) ? (bound << 16) | (unsigned int)(base + j) : 0xFFFFFFFFu;
                            best = min(best, key);
                        }
                    }
                }
                run = (int)uniform((unsigned int)wave_sum64((unsigned long long)run_l));
            } else {
                for (int i0 = 0; i0 < n_hot; i0 += 2 * kWave) {  // two slots per lane and step: two LDS round trips in flight, half the turns
                    const int ia = i0 + lane, ib = ia + kWave;
                    const bool ina = ia < n_hot, inb = ib < n_hot;
                    const unsigned int ma = ina ? d.load(ia) : 0u, mb = inb ? d.load(ib) : 0u;
                    const int ca = d.count(ma), cb = d.count(mb);
                    const bool prea = ina && ca >= d.L && ca < d.H, preb = inb && cb >= d.L && cb < d.H;
                    run += __popcll(wave_ballot(prea)) + __popcll(wave_ballot(preb));
                    const unsigned int ama = prea ? smask_of((unsigned int)ia) : 0u, amb = preb ? smask_of((unsigned int)ib) : 0u;
                    const int bounda = d.count(ma | ama), boundb = d.count(mb | amb);
                    const unsigned int keya = (prea && bounda < d.H) ? ((unsigned int)bounda << 16) | (unsigned int)ia : 0xFFFFFFFFu;
                    const unsigned int keyb = (preb && boundb < d.H) ? ((unsigned int)boundb << 16) | (unsigned int)ib : 0xFFFFFFFFu;
                    best = min(best, min(keya, keyb));
                }
            }
            s.running = run;
            running_exact = !owed;  // with nothing owed the state here is the reference's
            const unsigned int mn = wave_min32(best);
            ncand = 0;
            ci = 0;
            candv = 0u;
            if (mn != 0xFFFFFFFFu) {
                // the best slot, then the best slots of other lanes within one report of it
                unsigned long long pool = wave_ballot(best != 0xFFFFFFFFu && (best >> 16) <= (mn >> 16) + 1u);
                int src = __ffsll((long long)wave_ballot(best == mn)) - 1;
                for (int t = 0; t < kCand && src >= 0; ++t) {
                    const unsigned int v = (unsigned int)lane_value((int)best, src) & 0xFFFFu;
                    if (lane == t) candv = v;
                    ++ncand;
                    pool &= ~(1ull << src);
                    src = pool != 0ull ? __ffsll((long long)pool) - 1 : -1;
                }
            }
            set_witness();
            n_sweeps++;
            RAPID_T1(t_flush, ts0);
        };
        // the next candidate, after one sweep per window at most; false when none is left
        auto next_candidate = [&]() -> bool {
            ++ci;
            if (ci >= ncand) {
                if (swept) return false;
                swept = true;
                sweep();
            } else {
                set_witness();
            }
            return witness >= 0;
        };

        // the batch ends the cold / fast windows counted per lane -> s.batch
        auto fold_batches = [&]() {
            if (wave_ballot(vbatch != 0u) != 0ull) {
                s.batch += uniform((int)(unsigned int)wave_sum64((unsigned long long)vbatch));
                vbatch = 0u;
            }
        };
        // ---- apply the owed implicit reports.  Every report applied here was applied by the reference at a batch end
        // inside a window already certified emission-free (the witness of that window takes no uncounted implicit
        // report: its bound covers them all), so this cannot be where an emission happens.  Only at a batch boundary:
        // the applied state stands at one whenever something is owed (the carry holds the records after it).
        auto flush_pending = [&]() {
            if (!owed) return;
            owed = false;
            if (!s.seen_down) {  // the reference's pass is a no-op until a DOWN report has been seen (R/MultiNodeCutDetector.java:140-142)
                s.entered = true;
                return;
            }
            RAPID_TRACE_LINE("F r=%d\n", r);
            RAPID_T0(tf0);
            wave_lds_fence();
            int applied = 0;
            (void)invalidate_pairs(d, pairs, nullptr, nullptr, lane, &applied);
            s.entered = false;
            n_applied += applied;
            RAPID_T1(t_flush, tf0);
        };

        // ---- FAST window: a witness exists.  Certificate, evaluated BEFORE anything is applied: the witness -- in
        // preProposal since it was picked -- stays below H after the carried records and this window's records up to
        // its last batch end, even when credited with every implicit report it can ever get:
        // popc(state | the window's reports about it | wmask) < H.  Then it is in preProposal throughout,
        // updatesInProgress >= 1, and the reference cannot emit (R/MultiNodeCutDetector.java:110-121): the reports are
        // ORed into the slots in any order, nothing is returned, counted or rolled back.  Straight-line code, one
        // branch: nothing is applied unless the certificate holds, the last quarter holds a batch end (the records after
        // it are carried into the next window) and no first DOWN report would switch the implicit invalidation on inside
        // the window.
        auto fast_try = [&](const Win& cw) -> int {
            unsigned int so[kQ], w[kQ];
            // (everything that can stay in vector registers does: the CU's waves share ONE scalar pipe, and it is the busiest
            // unit of this kernel.  "Does the window say anything about the witness" is a running minimum of slot ^ witness,
            // looked at once; the batch ends are counted per lane by adding upper halves.)
            unsigned int wdiff = carry_so ^ witness_so;
            unsigned int nbv = 0u;  // batch ends in the window (none among the lanes that will be carried, by the choice of ncl)
            unsigned long long mEl;
            if constexpr (kLeanOpen) {
                // Boundary records of a vouched-for round, straight from the loaded dwords (no Rec in between): the alerts were
                // validated when the set was declared (index_touch_kernel: current configuration, a ring named, status against the
                // membership) and the deliveries are copies of them, so the only thing a delivered record can still be is LATE --
                // another configuration id: dropped whole, R/MembershipService.java:653-657.  A window taken here lies inside the
                // stream (its last window always goes the slow way), so no lane sees the zeros behind the end.  The batch end stays
                // out of the word that is applied (nobody reads it there): counted from the flags byte, and looked for in the last
                // quarter's flags.  Five vector instructions per quarter less than open() + effective().
#pragma unroll
                for (int q = 0; q < kQ; ++q) {
                    const Look k = lookup(cw.w3[q]);
                    const unsigned int raw = cw.w4[q];
                    bool current = true;
                    if constexpr (!kCurrent) current = (((unsigned long long)cw.c1[q] << 32) | (unsigned long long)cw.c0[q]) == cfg64;
                    const unsigned int full = (raw & kCoreRings) | ((raw & 0x00FF0000u) != 0u ? kCoreDown : kCoreUp);
                    w[q] = current ? full : 0u;
                    uncovered |= w[q] & k.entry;
                    so[q] = k.entry >> 16;
                    wdiff = min(wdiff, so[q] ^ witness_so);
                    nbv += (raw >> 24) & 1u;
                }
                mEl = wave_ballot((cw.w4[kQ - 1] & 0x01000000u) != 0u);
            } else {
                const Rec c = open(cw);
#pragma unroll
                for (int q = 0; q < kQ; ++q) {
                    const Look k = lookup(c.w3[q]);
                    w[q] = effective(c, q, k);
                    so[q] = k.entry >> 16;
                    wdiff = min(wdiff, so[q] ^ witness_so);
                    nbv += c.w4[q] >> 16;
                }
                mEl = wave_ballot((c.w4[kQ - 1] & kCoreEob) != 0u);
            }
            const unsigned long long mW = wave_ballot(wdiff == 0u);
            const int ncl = kWave - __clzll((long long)mEl);  // lanes of the last quarter up to its last batch end (0: none)
            const bool inl = lane < ncl;
            unsigned int wadd = 0u;  // what the window reports about the witness (rare: ten reports in a whole stream)
            if (mW != 0ull) {
                unsigned int acc = carry_so == witness_so ? carry_w : 0u;
#pragma unroll
                for (int q = 0; q < kQ; ++q) acc |= (so[q] == witness_so && (q < kQ - 1 || inl)) ? w[q] : 0u;
                // (one or two lanes hold something: picked out by name instead of a six-step butterfly over the wave)
                for (unsigned long long m = wave_ballot(acc != 0u); m != 0ull; m &= m - 1ull)
                    wadd |= (unsigned int)lane_value((int)acc, __ffsll((long long)m) - 1);
            }
            bool fastable = mEl != 0ull;
            if (!s.seen_down) {
                unsigned int any = carry_w;
#pragma unroll
                for (int q = 0; q < kQ; ++q) any |= w[q];
                fastable = fastable && wave_ballot((any & kCoreDown) != 0u) == 0ull;
            }
            const bool certified = __popc((wstate | wadd | wmask) & d.kmask) < d.H;
            if (!(fastable && certified)) {
                RAPID_TRACE_LINE("W-fail r=%d witness=%d wcount=%d fastable=%d\n", r, witness, __popc((wstate | wadd) & d.kmask), (int)fastable);
                return fastable ? kWitnessFails : kNotFastable;
            }
            // whole core words are ORed in: the status and batch-end bits land in bits of the state word nobody reads
            d.fast_or(carry_so, carry_w);
#pragma unroll
            for (int q = 0; q < kQ - 1; ++q) d.fast_or(so[q], w[q]);
            d.fast_or(so[kQ - 1], inl ? w[kQ - 1] : 0u);
            vbatch += nbv;
            carry_so = so[kQ - 1];
            carry_w = inl ? 0u : w[kQ - 1];
            carry_start = ncl;
            wstate |= wadd;
            owed = true;
            running_exact = false;
            return kApplied;
        };

        // ---- COLD window: no subject has reached L yet.  Applied with return values and kept iff no subject it touches
        // can reach H even when credited with every implicit report it can ever get: popc(old | reports | smask) < H.
        // Untouched subjects stay below L and receive nothing (implicit reports only go to subjects in preProposal), so
        // nothing crosses H anywhere in the window and the reference cannot emit.  The window's entrants (each crossing of
        // L is seen by exactly one lane, whatever the order of the atomics) are in preProposal from here on with a bound
        // below H: the first witnesses.
        auto cold_window = [&](const Win& cw) -> bool {
            const Rec c = open(cw);
            const unsigned long long mEl = wave_ballot((c.w4[kQ - 1] & kCoreEob) != 0u);
            if (mEl == 0ull) return false;
            const int ncl = kWave - __clzll((long long)mEl);
            const bool inl = lane < ncl;
            Dec e[kQ];
            unsigned int rb[kQ], old[kQ];
#pragma unroll
            for (int q = 0; q < kQ; ++q) {
                e[q] = decode_rec(c, q);
                rb[q] = e[q].slot < (unsigned int)n_hot ? e[q].bits : 0u;
            }
            const unsigned int rb_last = rb[kQ - 1];
            rb[kQ - 1] = inl ? rb_last : 0u;
            const unsigned int cs = carry_so >> 1, cb = cs < (unsigned int)n_hot ? (carry_w & d.kmask) : 0u;
            unsigned int oldc = 0u;
            if (cb != 0u) oldc = d.or_bits((int)cs, cb);
#pragma unroll
            for (int q = 0; q < kQ; ++q) {
                old[q] = 0u;
                if (rb[q] != 0u) old[q] = d.or_bits((int)e[q].slot, rb[q]);
            }
            bool high = cb != 0u && d.count(oldc | cb | smask_of(cs)) >= d.H;
#pragma unroll
            for (int q = 0; q < kQ; ++q) high = high || (rb[q] != 0u && d.count(old[q] | rb[q] | smask_of(e[q].slot)) >= d.H);
            if (wave_ballot(high) != 0ull) {  // roll back: every lane clears exactly the bits it set
                if ((cb & ~oldc) != 0u) d.clear_bits((int)cs, cb & ~oldc);
#pragma unroll
                for (int q = 0; q < kQ; ++q)
                    if ((rb[q] & ~old[q]) != 0u) d.clear_bits((int)e[q].slot, rb[q] & ~old[q]);
                wave_lds_fence();
                return false;
            }
            // entrants -> witness candidates
            unsigned int ent = 0xFFFFFFFFu;  // this lane's entrant slot, if any
            if (cb != 0u && d.count(oldc) < d.L && d.count(oldc | cb) >= d.L) ent = cs;
#pragma unroll
            for (int q = 0; q < kQ; ++q)
                if (rb[q] != 0u && d.count(old[q]) < d.L && d.count(old[q] | rb[q]) >= d.L) ent = e[q].slot;
            unsigned long long mEnt = wave_ballot(ent != 0xFFFFFFFFu);
            bool anyd = (carry_w & kCoreDown) != 0u;
#pragma unroll
            for (int q = 0; q < kQ; ++q) anyd = anyd || (e[q].down && (q < kQ - 1 || inl));
            if (!s.seen_down) s.seen_down = wave_ballot(anyd) != 0ull;
#pragma unroll
            for (int q = 0; q < kQ; ++q) vbatch += c.w4[q] >> 16;
            carry_so = 2u * e[kQ - 1].slot;
            carry_w = inl ? 0u : (e[kQ - 1].bits | (e[kQ - 1].down ? kCoreDown : 0u));
            carry_start = ncl;
            if (mEnt != 0ull) {
                cold = false;
                ncand = 0;
                ci = 0;
                candv = 0u;
                for (int t = 0; t < kCand && mEnt != 0ull; ++t) {
                    const unsigned int v = (unsigned int)lane_value((int)ent, __ffsll((long long)mEnt) - 1);
                    if (lane == t) candv = v;
                    ++ncand;
                    mEnt &= mEnt - 1ull;
                }
                set_witness();
                owed = true;
                running_exact = false;
            }
            return true;
        };

        // ---- SLOW path: the sub-chunk of <= careful_cap decoded records at scratch[spos ..) ----
        auto decode_scratch = [&]() {
            const int navail = min(careful_cap, send - spos);
            const unsigned int e = lane < navail ? scratch[spos + lane] : 0u;
            eob = (e & kDecEob) != 0u;
            // a sub-chunk ends at its last batch end (if it has one): no record is applied before the batch end
            // that precedes it has been processed
            mE_all = wave_ballot(eob);
            lastE = mE_all ? 63 - __clzll((long long)mE_all) : -1;
            ncons = lastE >= 0 ? lastE + 1 : navail;
            const bool valid = lane < ncons;
            eob = eob && valid;
            down = valid && (e & kDecDown) != 0u;
            bits = valid ? (e & 0x3FFFu) : 0u;
            dst = (int)((e >> 14) & kSlotMask);
        };
        // CAREFUL, first attempt: order-free application with the implicit invalidation applied immediately and an
        // EXACT count of the H crossings; rolled back if an emission cannot be excluded.  Returns false when the
        // sub-chunk must be narrowed or replayed record by record.
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
            const bool entered = s.entered || nLc > 0;
            const bool run_inv = lastE >= 0 && seen && entered;
            int nHi = 0, n_undo = 0, applied_here = 0;
            if (run_inv) {
                wave_lds_fence();
                nHi = invalidate_pairs(d, pairs, undo, &n_undo, lane, &applied_here);
            }
            const int Htot = nHc + nHi;
            if (Htot == 0 || s.running - Htot >= 1) {
                s.running += nLc - Htot;
                s.seen_down = seen;
                s.batch += __popcll(mE_all);
                s.entered = run_inv ? false : entered;
                if (nLc > 0) cold = false;
                n_applied += applied_here;
                n_records += (unsigned long long)ncons;
                spos += ncons;
                return true;
            }
            // Which record is critical: the one with the running-th explicit H crossing (the records before it hold fewer
            // crossings than updatesInProgress, and without an entrant no implicit report can be generated before it).
            // Only a hint for the size of the next attempt: every attempt is certified or replayed exactly like this one.
            hint_cap = -1;
            if (nHi == 0 && nLc == 0 && s.running >= 1 && s.running <= 16 && nHc >= s.running) {
                unsigned long long m = mH;
                for (int i = 1; i < s.running; ++i) m &= m - 1ull;
                hint_cap = __ffsll((long long)m) - 1;
            }
            if (n_undo > kUndoCap) {
                restart = true;  // cannot roll back: redo this receiver on the exact path only
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
        // EXACT path for the decoded sub-chunk: record by record
        auto exact_subchunk = [&]() {
            n_slow++;
            n_records += (unsigned long long)ncons;
            // only the records that do something: a report about a hot subject, a DOWN report, or a batch end
            for (unsigned long long todo = wave_ballot((lane < ncons) & ((bits != 0u) | (down & !s.seen_down) | eob)); todo != 0ull;
                 todo &= todo - 1ull) {
                const int q = __ffsll((long long)todo) - 1;
                const int qdst = lane_value(dst, q);
                const unsigned int qbits = (unsigned)lane_value((int)bits, q);
                const int qflags = lane_value((int)down | ((int)eob << 1), q);
                if (qflags & 1) s.seen_down = true;  // R/MultiNodeCutDetector.java:89-91, hot subject or not
                exact_apply(d, s, qdst, qbits, (qflags & 1) != 0, lane, nullptr, 0, nullptr);
                if (qflags & 2) {
                    exact_batch_end(d, s, pairs, lane, &n_applied);
                    if (s.batch_emitted) {  // R/MembershipService.java:333-335
                        emit_batch = s.batch;
                        break;
                    }
                    s.batch++;
                }
            }
            if (s.entered || s.running > 0) cold = false;
            spos += ncons;
        };
        // ---- SLOW window: the carried records and the window, decoded into the scratch list, go through the careful and
        // the exact path.  Everything is consumed: no carry afterwards.
        auto slow_window = [&](const Win& cw, int w) {
            const Rec c = open(cw);
            flush_pending();
            fold_batches();
            const int base_rec = w * kWin;
            Dec cr;
            cr.slot = carry_so >> 1;
            cr.bits = carry_w & d.kmask;
            cr.down = (carry_w & kCoreDown) != 0u;
            scratch[lane] = pack_rec(cr, false);
#pragma unroll
            for (int q = 0; q < kQ; ++q) {
                // the stream's last record closes its batch whatever its flag says
                const bool e_eob = (c.w4[q] & kCoreEob) != 0u || base_rec + q * kWave + lane == nrec - 1;
                scratch[kWave + q * kWave + lane] = pack_rec(decode_rec(c, q), e_eob);
            }
            spos = carry_start;
            send = kWave + min(kWin, nrec - base_rec);
            carry_w = 0u;
            carry_start = kWave;
            wave_lds_fence();
            if (!running_exact) sweep();  // updatesInProgress, exactly (nothing is owed any more)
            RAPID_T0(tc0);
            bool exact_next = false;  // the sub-chunk starts with the critical record: replay it record by record
            while (spos < send && emit_batch < 0 && !restart) {
                n_careful++;
                RAPID_TRACE_LINE("C r=%d w=%d spos=%d run=%d cap=%d\n", r, w, spos, s.running, careful_cap);
                decode_scratch();
                if (exact_only || s.batch_emitted || exact_next) {
                    exact_next = false;
                    exact_subchunk();
                    if (!exact_only && !s.batch_emitted) careful_cap = kWave;
                } else if (immediate_subchunk()) {
                    careful_cap = min(kWave, careful_cap * 2);
                } else if (!restart) {
                    // An emission cannot be excluded somewhere in these ncons records.  Narrow the sub-chunk instead of
                    // replaying all of them one by one: the record-by-record path only ever runs on a few records.
                    if (hint_cap >= 1 && hint_cap < ncons) {  // the records before the critical crossing
                        careful_cap = hint_cap;
                        continue;
                    }
                    if (hint_cap == 0 && ncons > 2) {  // the first record is the critical one
                        careful_cap = 2;
                        exact_next = true;
                        continue;
                    }
                    if (ncons > 4) {
                        careful_cap = ncons / 2;
                        continue;  // same position, smaller sub-chunk
                    }
                    exact_subchunk();
                    careful_cap = kWave;
                }
            }
            need_sweep = true;
            RAPID_T1(t_careful, tc0);
        };

        for (;;) {  // one pass over the stream; a second one, on the exact path only, after an undo-list overflow
            // ---- detector state: nothing reported yet ----
            uint4* st = reinterpret_cast<uint4*>(mine);
            for (int i = lane; i < state_bytes / 16; i += kWave) st[i] = make_uint4(0, 0, 0, 0);
            s.running = 0;
            s.batch = 0;
            s.proposal_count = 0;
            s.seen_down = false;
            s.entered = false;
            s.batch_emitted = false;
            cold = true;
            owed = false;
            running_exact = true;
            need_sweep = false;
            witness = -1;
            ncand = 0;
            ci = 0;
            carry_so = 0u;
            carry_w = 0u;
            carry_start = kWave;
            vbatch = 0u;
            careful_cap = kWave;
            if (restart) {  // the stream again, from its first window
                restart = false;
                exact_only = true;
#pragma unroll
                for (int i = 0; i < kSets; ++i) load_window(rsrc, lane_off + (unsigned int)i * kWinBytes, S[i]);
            }
            wave_lds_fence();
            unsigned int voff = lane_off + (unsigned int)kSets * kWinBytes;  // this lane's offset in the next window to request
            for (int w = 0; w < nwin && emit_batch < 0 && !restart; ++w) {
                // ---- steady state: a run of fast windows, in a loop of its own that holds nothing but what a fast window
                // needs (the general iteration below carries the whole receiver's bookkeeping through every window and costs
                // several times the window's own instructions in copies and spilled scalars).  It ends at the window that
                // owes something else -- the claim of the next receiver, the stream's last window -- or at the first
                // window that cannot be certified, which the general iteration takes over untouched.
                swept = false;
                const bool lean_ok = !exact_only && !s.batch_emitted && (p.flags & (8 | 32)) == 0;
                if (lean_ok && need_sweep && !cold && w + 1 < nwin) {  // (after a slow window: updatesInProgress and fresh witness candidates)
                    need_sweep = false;
                    swept = true;
                    sweep();
                }
#ifndef RAPID_COLD_IN_TURN
#define RAPID_COLD_IN_TURN 1
#endif
                constexpr bool kColdInTurn = RAPID_COLD_IN_TURN != 0;  // (0: a stream's first, cold windows go through the general iteration)
                if (lean_ok && !need_sweep && ((kColdInTurn && cold) || (!cold && witness >= 0))) {
                    const int w_end = nwin - 1;  // the stream's last window goes the slow way
                    const int w_first = w;
                    RAPID_T0(tl0);
                    // A TURN takes kSets windows, each from its own registers, and requests each set again as soon as its step is
                    // over: the next window of that set if the step applied its window, the SAME window again if it could not
                    // certify it or was skipped because an earlier step of the turn failed.  Every step thus ends with a load
                    // into its set on every path -- there is no "reloaded or not" for the compiler to merge with a copy (a copy of
                    // a register that is still in flight costs an s_waitcnt vmcnt(0) per turn: measured, the whole memory
                    // latency exposed) -- at the price of reading a window twice after a failed certificate (a few per receiver).
                    unsigned int vturn = voff - (unsigned int)kSets * kWinBytes;  // this lane's offset in the window held by S[0]
                    int n_ok = kSets;
                    // (the last turn before the stream's final window is a partial one: its remaining steps are skipped like the
                    // steps behind a failed certificate, which re-requests the final window -- a kilobyte or two per receiver --
                    // and saves the general iterations, with their copies and their wait, that the tail would otherwise take)
                    while (w < w_end && n_ok == kSets) {
                        if (!claimed && w + kSets + kClaimAhead > nwin) claim();  // (a few windows early rather than in the middle of a turn)
                        n_ok = 0;
#pragma unroll
                        for (int i = 0; i < kSets; ++i) {
                            bool advance = false;
                            if (n_ok == i && w + i < w_end) {
                                if (kColdInTurn && cold) {  // (the first windows of a stream: until a subject reaches L there is no witness to be had)
                                    advance = cold_window(S[i]);
                                } else if (witness >= 0) {
                                    int st_ = fast_try(S[i]);
                                    // a witness that fails is replaced right here while candidates from the last sweep are left
                                    // (a new sweep is the general iteration's business).  The first attempt stands alone: as the
                                    // head of a retry loop it would start with a block of copies for everything the loop carries.
                                    if (__builtin_expect(st_ == kWitnessFails, 0)) {
                                        while (st_ == kWitnessFails && ci + 1 < ncand) {
                                            ++ci;
                                            set_witness();
                                            st_ = fast_try(S[i]);
                                        }
                                    }
                                    advance = st_ == kApplied;
                                }
                            }
                            if (advance) ++n_ok;
                            stream_settle(uncovered, carry_w, carry_so);
                            load_window(rsrc, vturn + (unsigned int)i * kWinBytes + (advance ? (unsigned int)kSets * kWinBytes : 0u), S[i]);
                        }
                        w += n_ok;
                        vturn += (unsigned int)n_ok * kWinBytes;  // (a complete turn: all sets moved on; else: see below)
                    }
                    // After a turn that stopped at step i = n_ok: S[0 .. i-1] hold the windows w + kSets - i .., S[i ..] the windows
                    // w .. (re-requested).  Back into stream order, S[0] = window w (rare; the copies wait for the data).
                    const int failed = n_ok == kSets ? 0 : n_ok;
#pragma unroll
                    for (int t_ = 1; t_ < kSets; ++t_) {
                        if (t_ <= failed) {  // one place to the left, `failed` times
                            const Win t = S[0];
#pragma unroll
                            for (int i = 0; i + 1 < kSets; ++i) S[i] = S[i + 1];
                            S[kSets - 1] = t;
                        }
                    }
                    voff = vturn + (unsigned int)kSets * kWinBytes;
                    RAPID_T1(t_ensure, tl0);
                    RAPID_HOOK_TIGHT(w - w_first);
                    n_fast += (unsigned long long)(w - w_first);
                    n_records += (unsigned long long)(w - w_first) * (unsigned long long)kWin;
                    if (w != w_first) swept = false;  // (another window now)
                }
                // ---- the general iteration: window w, in place in S[0].  The sets move up one place and the next window is requested
                // AFTER the window has been dealt with: the copies need the youngest request to have landed, and what this
                // iteration does (a cold window, a sweep, a slow window) is time that request has had by then.
                const Win& cur = S[0];
                auto next_general = [&]() {
                    stream_settle(uncovered, carry_w, carry_so);
#pragma unroll
                    for (int i = 0; i + 1 < kSets; ++i) S[i] = S[i + 1];
                    load_window(rsrc, voff, S[kSets - 1]);  // kSets windows ahead
                    voff += kWinBytes;
                };
                if (!claimed && w + kClaimAhead >= nwin) claim();
                if ((p.flags & 32) != 0) {  // measurement aid: stream the records through the registers without tallying them
#pragma unroll
                    for (int q = 0; q < kQ; ++q) {
                        sink ^= cur.w3[q] ^ cur.w4[q];
                        if constexpr (kFmt == kFmtBoundary && !kCurrent) sink ^= cur.c0[q] ^ cur.c1[q];
                    }
                    next_general();
                    continue;
                }
                bool done = false;
                // the stream's last window always goes the slow way (its last record closes a batch without saying so)
                if (!exact_only && !s.batch_emitted && (p.flags & 8) == 0 && w + 1 < nwin) {
                    RAPID_T0(tl0);
                    if (need_sweep && !cold) {
                        need_sweep = false;
                        swept = true;
                        sweep();
                    }
                    if (cold) {
                        done = cold_window(cur);
                    } else if (witness >= 0) {
                        int st_ = fast_try(cur);
                        while (st_ == kWitnessFails && next_candidate()) st_ = fast_try(cur);
                        done = st_ == kApplied;
                    }
                    if (done) {
                        n_fast++;
                        n_records += (unsigned long long)kWin;
                    } else {
                        n_pipe++;
                    }
                    RAPID_T1(t_lean, tl0);
                }
                if (!done) slow_window(cur, w);
                next_general();
            }
            if (!restart) break;
        }
        RAPID_T0(to0);
        if ((p.flags & 4) != 0) {  // measurement aid: ~35 k idle cycles between two receivers (what C3b's end phase costs)
            for (int i = 0; i < 4; ++i) __builtin_amdgcn_s_sleep(127);
        }
        if (!claimed) claim();
        if (r_next >= p.n_receivers && p.pool != nullptr) {
            // Own deal worked off: one receiver from the common pool.  Claimed here and not windows ahead like the others --
            // the answer of a global atomic lives in a vector register, and one that stays live across window iterations
            // costs the precise wait counts of the stream (stream_load.h); here nothing is in flight.  The round trip
            // (~2 us) is paid by the last eighth of the receivers only.
            unsigned int g = 0u;
            if (lane == 0) g = atomicAdd(p.pool, 1u);
            const unsigned int gi = uniform(g);
            if (gi < (unsigned int)(p.n_receivers - p.n_static)) {
                r_next = p.n_static + (int)gi;
                next0 = stream_scalar_load(p.rec_off + r_next);
                next1 = stream_scalar_load(p.rec_off + r_next + 1);
            }
        }
        // start the next receiver's stream now; its first window lands while this receiver's results are written
        int nrec_next = 0;
        if (r_next < p.n_receivers) {
            const long long rec0 = next0, rec1 = next1;
            nrec_next = checked_length(rec0, rec1);
            rsrc = make_stream(rec0, rec0 + nrec_next);
#pragma unroll
            for (int i = 0; i < kSets; ++i) load_window(rsrc, lane_off + (unsigned int)i * kWinBytes, S[i]);
        }

        // ---- outputs: the proposal = every flushed (hot) slot, ascending node index ----
        int count = 0;
        unsigned long long fp = 0;
        if (emit_batch >= 0) {
            wave_lds_fence();
            int* const out = p.props + (long long)r * p.prop_cap;
            unsigned long long* const bm = p.bitmaps != nullptr ? p.bitmaps + (long long)r * p.bitmap_words : nullptr;
            if constexpr (nos_in_lds) {
                for (int i0 = 0; i0 < n_hot; i0 += kWave) {
                    const int i = i0 + lane;
                    const bool take = i < n_hot && (d.load(i) & Det::kFlushed) != 0;
                    const unsigned long long mk = wave_ballot(take);
                    if (bm != nullptr && lane == 0) stream_store(bm + (i0 >> 6), mk);
                    const int idx = count + __popcll(mk & lanes_lt(lane));
                    if (take) {
                        const int node = l_nos[i];
                        if (idx < p.prop_cap) stream_store(out + idx, node);
                        fp += mix64((unsigned long long)node);
                    }
                    count += __popcll(mk);
                }
            } else {
                // slot -> node in memory (packed rounds): the nodes of eight steps are requested together, flushed or not -- coalesced
                // 256-byte loads, all in flight at once -- and the NEXT eight steps' before these eight are written out.  Asked for
                // under the "flushed" mask, every step waited out a memory round trip of its own: 235 of them per proposal at 15,000
                // hot subjects (a tenth of a receiver's time at 10^6 nodes, profiles/r06_c5_phase_resolved.txt).
                constexpr int kOutBatch = 8;
                auto fetch = [&](int i0, int (&nodes)[kOutBatch]) {
#pragma unroll
                    for (int u = 0; u < kOutBatch; ++u) {
                        // (clamped, not skipped: a load under a branch leaves the compiler's wait-count pass without a count, and it
                        // then waits with vmcnt(0) -- for the loads just requested and for every store still on its way)
                        const int i = min(i0 + u * kWave + lane, n_hot - 1);
                        nodes[u] = p.idx.node_of_slot[i];
                    }
                };
                auto emit = [&](int i0, const int (&nodes)[kOutBatch]) {
#pragma unroll
                    for (int u = 0; u < kOutBatch; ++u) {
                        const int i = i0 + u * kWave + lane;
                        if (i0 + u * kWave >= n_hot) continue;  // (wave-uniform)
                        const bool take = i < n_hot && (d.load(i) & Det::kFlushed) != 0;
                        const unsigned long long mk = wave_ballot(take);
                        if (bm != nullptr && lane == 0) stream_store(bm + ((i0 + u * kWave) >> 6), mk);
                        const int idx = count + __popcll(mk & lanes_lt(lane));
                        if (take) {
                            if (idx < p.prop_cap) stream_store(out + idx, nodes[u]);
                            fp += mix64((unsigned long long)nodes[u]);
                        }
                        count += __popcll(mk);
                    }
                };
                constexpr int kStep = kWave * kOutBatch;
                int na[kOutBatch], nb[kOutBatch];
                fetch(0, na);
                for (int i0 = 0; i0 < n_hot; i0 += 2 * kStep) {
                    fetch(i0 + kStep, nb);
                    emit(i0, na);
                    fetch(i0 + 2 * kStep, na);
                    emit(i0 + kStep, nb);
                }
            }
            fp = wave_sum64(fp) + mix64(0x5EEDull + (unsigned long long)count);
            if (fp == 0) fp = 1;
        }
        if (wave_ballot(uncovered != 0u) != 0ull) {  // the declared alert set does not cover what was delivered: the results are void
            if (lane == 0) stream_flag_or(p.error_flags, 1u);
            uncovered = 0u;
        }
        if (lane == 0) {
            stream_store(p.emit_batch + r, emit_batch);
            stream_store(p.num_proposals + r, s.proposal_count);
            stream_store(p.prop_count + r, count > p.prop_cap ? -1 : count);
            stream_store(p.fingerprint + r, fp);
            if (count != 0 && p.vote_acc != nullptr) {  // this receiver votes (an oversized proposal is still a vote)
                atomicAdd(&block_votes[2], 1ull);
                atomicMax(&block_votes[3], ~(unsigned long long)(unsigned int)r);
                if (p.vote_cand != nullptr && (p.flags & 1024) == 0) {  // noted for the workgroup's own comparison at its end (TallyParams::vote_cand)
                    const unsigned int at = atomicAdd(&cand_flags[2], 1u);
                    if (at < (unsigned int)kBlockVoters) {
                        block_voters[at] = (unsigned int)r;
                    } else {  // (a workgroup that took far more than its share from the pool: the launch's last workgroup compares)
                        const unsigned int g = atomicAdd(p.vote_deferred, 1u);
                        if (g < (unsigned int)p.vote_deferred_cap) p.vote_deferred[1 + g] = (unsigned int)r;
                    }
                }
            }
            RAPID_HOOK_RESULTS();
        }
        wave_lds_fence();
        r = r_next;
        nrec = nrec_next;
        RAPID_T1(t_out, to0);
        RAPID_HOOK_RECEIVER_END();
    }
    unsigned long long mine_stats[8];
    mine_stats[0] = n_slow; mine_stats[1] = n_fast; mine_stats[2] = n_sweeps; mine_stats[3] = n_restart;
    mine_stats[4] = (unsigned long long)n_applied; mine_stats[5] = n_records; mine_stats[6] = n_pipe; mine_stats[7] = n_careful;
    RAPID_HOOK_STATS();
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
            if (!RAPID_HOOK_REDUCE(i)) atomicAdd(&block_stats[i], mine_stats[i]);
    }
    if ((p.flags & 32) != 0 && sink == 0x12345678u) block_stats[0] = 1ull;
    __syncthreads();
    if (threadIdx.x == 0 && p.pool != nullptr) {
        if (atomicAdd(&p.pool[1], 1u) == gridDim.x - 1u) {  // every other workgroup has made its last claim
            p.pool[0] = 0u;
            p.pool[1] = 0u;
        }
    }
    if (p.vote_cand != nullptr) {  // (kernel-uniform) this workgroup's voters against the round's candidate
        // Here, at the workgroup's end, and not where each receiver finishes: the waves have stopped streaming, so the fences and the
        // round trips to the candidate's words wait for nothing but themselves (done per receiver, inside the receiver loop, the same
        // comparison cost the launch 0.04 ms at C3b: every fence drains the next stream's windows in flight and drops the CU's L1).
        stream_drain();
        __syncthreads();  // every result this workgroup stored has left the CU
        // (No fence here: on gfx950 an agent-scope fence writes back and invalidates the XCD's whole L2 -- 3,840 waves doing that at the
        // end of a launch cost it 0.05 ms.  What other workgroups read of this one's results they read with agent-scope atomic loads,
        // after this workgroup's ONE release fence further down -- thread 0's, behind the barrier that drained everybody's stores.)
        const unsigned int nv_all = (p.flags & 2048) != 0 ? 0u : cand_flags[2];
        const unsigned int nv = nv_all < (unsigned int)kBlockVoters ? nv_all : (unsigned int)kBlockVoters;
        const int words = p.bitmap_words;
        if (nv != 0u) {
            if (threadIdx.x == 0) {
                const unsigned int r0 = block_voters[0];
                const unsigned int prev = atomicCAS(reinterpret_cast<unsigned int*>(p.vote_cand), 0u, r0 + 1u);
                if (prev == 0u) {  // nobody has voted before: this workgroup's first voter holds the round's CANDIDATE
                    for (int w = 0; w < words; ++w)
                        __hip_atomic_store(p.vote_cand + 4 + w, __hip_atomic_load(p.bitmaps + (long long)r0 * words + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(p.vote_cand + 2, __hip_atomic_load(p.fingerprint + r0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(p.vote_cand + 3, (unsigned long long)(unsigned int)__hip_atomic_load(p.prop_count + r0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __threadfence();  // (release: the candidate's words before the flag)
                    __hip_atomic_store(p.vote_cand + 1, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    cand_flags[0] = 1u;
                } else {  // (what is read of the candidate below is read with agent-scope atomic loads: no acquire fence)
                    const unsigned long long ready = __hip_atomic_load(p.vote_cand + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    cand_flags[0] = ready != 0ull ? 1u : 2u;
                }
            }
            __syncthreads();
            if (cand_flags[0] == 1u) {  // published: one thread per voter, the bitmaps compared word for word
                const unsigned long long cf = __hip_atomic_load(p.vote_cand + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                for (unsigned int i = threadIdx.x; i < nv; i += blockDim.x) {
                    const unsigned int rx = block_voters[i];
                    if (__hip_atomic_load(p.fingerprint + rx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != cf) continue;
                    const bool differs = bitmap_differs(p.bitmaps + (long long)rx * words, p.vote_cand + 4, words);
                    atomicAdd(&block_votes[1], 1ull);
                    if (differs) atomicAdd(&block_votes[0], 1ull);
                }
            } else {  // claimed by another workgroup a moment ago, not yet published: the launch's last workgroup compares these
                for (unsigned int i = threadIdx.x; i < nv; i += blockDim.x) {
                    const unsigned int g = atomicAdd(p.vote_deferred, 1u);
                    if (g < (unsigned int)p.vote_deferred_cap) p.vote_deferred[1 + g] = block_voters[i];
                }
            }
            __syncthreads();
        }
    }
    if (threadIdx.x == 0 && p.vote_acc != nullptr) {
        if (block_votes[2] != 0ull) {
            atomicAdd(&p.vote_acc[2], block_votes[2]);
            atomicMax(&p.vote_acc[3], block_votes[3]);
            if (block_votes[1] != 0ull) atomicAdd(&p.vote_acc[1], block_votes[1]);
            if (block_votes[0] != 0ull) atomicAdd(&p.vote_acc[0], block_votes[0]);
        }
        __threadfence();  // this workgroup's share is visible before its count is (release; once per workgroup)
        if (atomicAdd(&p.vote_acc[4], 1ull) == (unsigned long long)gridDim.x - 1ull) {  // the last workgroup: every other one has added its share
            __threadfence();  // (acquire)
            if (p.vote_cand != nullptr) {
                cand_flags[1] = 1u;  // (the answer is completed by the whole workgroup, below)
                block_votes[0] = 0ull;
                block_votes[1] = 0ull;
            } else {
                const unsigned long long voters = atomicAdd(&p.vote_acc[2], 0ull), repc = atomicAdd(&p.vote_acc[3], 0ull);
                p.vote_res[0] = voters != 0ull ? (~repc & 0xFFFFFFFFull) : 0xFFFFFFFFull;
                p.vote_res[1] = 0ull;
                p.vote_res[2] = voters;
                for (int i = 3; i < 10; ++i) p.vote_res[i] = 0ull;
                for (int i = 0; i < 5; ++i) p.vote_acc[i] = 0ull;
            }
        }
    }
    if (p.vote_cand != nullptr) {  // (kernel-uniform)
        __syncthreads();
        if (cand_flags[1] != 0u) {  // the launch's last workgroup: every other one has added its share and published what it wrote
            // the voters that finished between the candidate's claim and its publication (usually none): compared here, one thread
            // each, against the published candidate -- their bitmaps and fingerprints were written before their workgroups counted
            const int words = p.bitmap_words;
            const unsigned int n_def_all = (p.flags & 4096) != 0 ? 0u : __hip_atomic_load(p.vote_deferred, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned int n_def = n_def_all < (unsigned int)p.vote_deferred_cap ? n_def_all : (unsigned int)p.vote_deferred_cap;
            const unsigned long long cf = __hip_atomic_load(p.vote_cand + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // (everything the answer needs is requested here, together: one round trip instead of four at the very end of the launch)
            const unsigned long long owner1 = __hip_atomic_load(p.vote_cand, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0xFFFFFFFFull;
            const unsigned long long voters = __hip_atomic_load(p.vote_acc + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long acc_seen = __hip_atomic_load(p.vote_acc + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long acc_bad = __hip_atomic_load(p.vote_acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int cnt = (int)(unsigned int)__hip_atomic_load(p.vote_cand + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned int tally_err = __hip_atomic_load(p.error_flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int n_list = (owner1 == 0ull || cnt < 0 || cnt > p.prop_cap) ? 0 : cnt;
            const int* const src = p.props + (long long)(owner1 == 0ull ? 0ull : owner1 - 1ull) * p.prop_cap;
            int my_nodes[4];  // (the candidate's node list, up to four nodes per thread in flight: the whole list of a cut of 4,096)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = (int)threadIdx.x + j * (int)blockDim.x;
                my_nodes[j] = i < n_list ? __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
            }
            for (unsigned int i = threadIdx.x; i < n_def; i += blockDim.x) {
                const unsigned int rx = __hip_atomic_load(p.vote_deferred + 1 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__hip_atomic_load(p.fingerprint + rx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != cf) continue;
                const bool differs = bitmap_differs(p.bitmaps + (long long)rx * words, p.vote_cand + 4, words);
                atomicAdd(&block_votes[1], 1ull);
                if (differs) atomicAdd(&block_votes[0], 1ull);
            }
            __syncthreads();
            unsigned long long seen = acc_seen + block_votes[1];
            const unsigned long long bad = acc_bad + block_votes[0];
            const bool lost = n_def_all > n_def;  // more deferred voters than the list holds: their votes are unknown -- no quorum is claimed
            if (lost) seen = 0ull;
            unsigned long long res[10];
            res[0] = owner1 != 0ull ? owner1 - 1ull : 0xFFFFFFFFull;
            res[1] = seen;
            res[2] = voters;
            res[3] = voters == 0ull ? 0ull : (seen == voters ? 1ull : 2ull);
            res[4] = owner1 != 0ull ? cf : 0ull;
            res[5] = owner1 != 0ull ? ~cf : ~0ull;
            res[6] = bad;
            res[7] = seen;
            res[8] = (unsigned long long)tally_err;
            res[9] = 0ull;
            __syncthreads();  // (everybody has read the accumulators: thread 11 zeroes them below)
            // the answer block: res[0..9], then {size, node list} of the candidate -- in memory (the all-gather of a sharded population
            // reads it there) and, for a population held by one rank, in the host-mapped page the host polls
            int* const ref = reinterpret_cast<int*>(p.vote_res + 10);
            unsigned long long* const pub = const_cast<unsigned long long*>(p.vote_publish);
            int* const pref = pub != nullptr ? reinterpret_cast<int*>(pub + 10) : nullptr;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = (int)threadIdx.x + j * (int)blockDim.x;
                if (i < n_list) {
                    ref[1 + i] = my_nodes[j];
                    if (pref != nullptr) pref[1 + i] = my_nodes[j];
                }
            }
            for (int i = (int)threadIdx.x + 4 * (int)blockDim.x; i < n_list; i += (int)blockDim.x) {
                const int node = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ref[1 + i] = node;
                if (pref != nullptr) pref[1 + i] = node;
            }
            if (threadIdx.x < 10u) {
                p.vote_res[threadIdx.x] = res[threadIdx.x];
                if (pub != nullptr) pub[threadIdx.x] = res[threadIdx.x];
            }
            if (threadIdx.x == 10u) {
                ref[0] = owner1 != 0ull ? cnt : 0;
                if (pref != nullptr) pref[0] = owner1 != 0ull ? cnt : 0;
            }
            if (threadIdx.x == 11u) {  // everything a launch expects zeroed, for the next one
                for (int i = 0; i < 5; ++i) p.vote_acc[i] = 0ull;
                for (int i = 0; i < 4; ++i) p.vote_cand[i] = 0ull;
                p.vote_deferred[0] = 0u;
            }
            if (pub != nullptr) {
                __syncthreads();  // (everybody's stores to the page have left the CU)
                if (threadIdx.x == 0) {
                    __threadfence_system();  // ONE system-scope fence (each one writes back the L2), then the word the host polls
                    if (p.vote_seq_out != nullptr) *p.vote_seq_out = p.vote_seq;
                }
            }
        }
    }
    // p.stats = [gridDim.x][8], accumulated over launches; one plain read-modify-write per workgroup and counter
    if (threadIdx.x < 8u && p.stats != nullptr && !RAPID_HOOK_STATS_OUT(threadIdx.x)) p.stats[(size_t)blockIdx.x * 8 + threadIdx.x] += block_stats[threadIdx.x];
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
    s.entered = false;
    s.batch = 0;
    s.batch_emitted = false;
    int total = 0;
    if (p.mode == 0) {
        for (int a = 0; a < p.n_alerts; ++a) {
            const unsigned int* w = reinterpret_cast<const unsigned int*>(p.alerts + (long long)a * kRecBytes);
            const int dst = uniform((int)w[3]);
            const unsigned int w4 = uniform(w[4]);
            const int before = total;
            if ((unsigned)dst < (unsigned)p.n_nodes)
                exact_apply(d, s, dst, w4 & d.kmask, ((w4 >> 16) & 0xFFu) != 0, lane, p.out_idx, p.out_cap, &total);
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

#ifdef RAPID_TEST_BUILD
#include "stream_probes.inc"  // (stream_probe_kernel, dma_probe_kernel: measurement aids, test build only)
#endif  // RAPID_TEST_BUILD

}  // namespace rapid
