/*
 * rapid_mi355x.h -- C ABI of the MI355X-native cut-detection / consensus-counting engine.
 *
 * This is the drop-in boundary for ONE path of lalithsuresh/rapid: the package-private Java classes
 * MembershipView, MultiNodeCutDetector and the fast round of FastPaxos, plus the per-batch semantics of
 * MembershipService.handleMessage(BatchedAlertMessage), replayed for a whole simulated population on the GPU;
 * and, host only, what sits either side of that path: the wire forms of its messages and the consensus
 * instance of one node (FastPaxos + Paxos) for the rounds the fast path does not decide.
 * Citations are relative to /root/reference/rapid/src/main/java/com/vrg/rapid/ ("R/").
 *
 * Conventions:
 *   - plain C, opaque handles, no C++ types, no exceptions across the boundary;
 *   - endpoints are dense int32 node indices assigned by the host (members AND known joiners);
 *   - all inputs are borrowed for the duration of the call; all outputs go to caller-allocated buffers with
 *     an explicit capacity and an out-count; the library never returns memory it owns;
 *   - every call returns 0 (RAPID_OK) or a negative code that maps 1:1 onto the Java exception the
 *     reference would throw; rapid_last_error() returns a human-readable detail;
 *   - threading: one caller at a time per handle, re-entrant across handles (the reference confines these
 *     classes to its single "protocol" executor thread, R/SharedResources.java:53); no callbacks into the host.
 *   - the library fails loudly (RAPID_EDEVICE) when no gfx950 device is usable -- there is no CPU fallback.
 */
#ifndef RAPID_MI355X_H
#define RAPID_MI355X_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- error codes ------------------------------------------------------------------------------------- */
#define RAPID_OK 0
#define RAPID_EINVAL (-1)        /* IllegalArgumentException          R/MultiNodeCutDetector.java:52-55      */
#define RAPID_ENODE_EXISTS (-2)  /* NodeAlreadyInRingException        R/MembershipView.java:133-135, 502-506 */
#define RAPID_ENODE_MISSING (-3) /* NodeNotInRingException            R/MembershipView.java:172-174, 508-512 */
#define RAPID_EUUID_SEEN (-4)    /* UUIDAlreadySeenException          R/MembershipView.java:127-129, 514-519 */
#define RAPID_ECAPACITY (-5)     /* a caller buffer / configured capacity is too small                        */
#define RAPID_EDEVICE (-6)       /* HIP / RCCL failure, or no gfx950 device                                   */
#define RAPID_ESTATE (-7)        /* call made in the wrong order (e.g. tally before streams were loaded)      */
#define RAPID_ECOLLISION (-8)    /* two different proposals shared a fingerprint (verified, never silent)     */

/* ---- JoinStatusCode (rapid/src/main/proto/rapid.proto:84-90) ------------------------------------------ */
#define RAPID_HOSTNAME_ALREADY_IN_RING 0
#define RAPID_UUID_ALREADY_IN_RING 1
#define RAPID_SAFE_TO_JOIN 2

/* ---- EdgeStatus (rapid.proto:114-117) ---------------------------------------------------------------- */
#define RAPID_EDGE_UP 0
#define RAPID_EDGE_DOWN 1

/* ---- packed alert record: the device-resident form of AlertMessage (rapid.proto:102-111) ---------------
 * 20 bytes, 4-byte aligned.  ring_mask bit k <=> ringNumber k (0 <= k < K <= 14); one record = one
 * AlertMessage whose ringNumber list is the set bits in ascending order (as getRingNumbers produces them,
 * R/MembershipView.java:397-418).  The joiner's NodeId of an UP alert is the id registered for node `dst`.
 * flags bit0 marks the last record of its BatchedAlertMessage (rapid.proto:95-99); the end of a receiver's
 * stream also closes a batch. */
#pragma pack(push, 1)
typedef struct rapid_alert_record {
    int64_t cfg_id;     /* AlertMessage.configurationId */
    uint32_t src;       /* edgeSrc (stored, never read by the detector -- R/MultiNodeCutDetector.java:101) */
    uint32_t dst;       /* edgeDst */
    uint16_t ring_mask; /* ringNumber[] */
    uint8_t status;     /* RAPID_EDGE_UP / RAPID_EDGE_DOWN */
    uint8_t flags;      /* RAPID_ALERT_LAST_IN_BATCH */
} rapid_alert_record;
#pragma pack(pop)
#define RAPID_ALERT_LAST_IN_BATCH 1u
#define RAPID_MAX_K 14

typedef struct rapid_engine rapid_engine; /* one simulated cluster on one GPU */
typedef struct rapid_cd rapid_cd;         /* one MultiNodeCutDetector instance (device-resident state) */
typedef struct rapid_fast_round rapid_fast_round; /* one FastPaxos fast-round vote counter */

typedef struct rapid_engine_config {
    int32_t n_max;     /* capacity in node indices (members + joiners) */
    int32_t K;         /* rings; 3 <= K <= RAPID_MAX_K        (R/MultiNodeCutDetector.java:39, R/Cluster.java:72) */
    int32_t H;         /* high watermark, L <= H <= K         (R/Cluster.java:73) */
    int32_t L;         /* low watermark, 1 <= L               (R/Cluster.java:74) */
    int32_t device_id; /* HIP device ordinal */
    int32_t max_cut;   /* capacity of one receiver's proposal list; 0 = default (min(n_max, 4096)) */
} rapid_engine_config;

/* ---- lifecycle ----------------------------------------------------------------------------------------
 * rapid_engine_create validates K/H/L exactly like the MultiNodeCutDetector constructor
 * (R/MultiNodeCutDetector.java:51-55; constructed at R/Cluster.java:265-267 and :458-460). */
int rapid_engine_create(const rapid_engine_config* cfg, rapid_engine** out);
void rapid_engine_destroy(rapid_engine* h);
const char* rapid_last_error(const rapid_engine* h);
int rapid_device_count(void); /* number of usable gfx950 devices (0 on a CPU-only host) */
/* Known-answer test of the whole path on h's device, on a private engine (h's view and streams are untouched): a 5-node view whose
 * configuration id and ring 0 are compared with constants (the CPU oracle's values, recomputed by tests/test_abi.py), one round in
 * which the four other receivers hear node 4 reported DOWN on every ring, the decided cut {4} and the configuration id after it.
 * RAPID_OK, or RAPID_EDEVICE with the first difference in rapid_last_error(h).  What a host calls once after rapid_engine_create,
 * where the reference would simply trust `new MembershipView(K, ids, endpoints)` (R/MembershipView.java:74-89) and
 * `new MultiNodeCutDetector(K, H, L)` (R/MultiNodeCutDetector.java:51-60) not to fail: a runtime that cannot run the kernels is
 * then an error code at start-up, not a wrong cut later (a device FAULT still ends the process: INTEGRATION.md section 6). */
int rapid_engine_self_test(rapid_engine* h);

/* ---- MembershipView (R/MembershipView.java) -----------------------------------------------------------
 * rapid_view_build <- MembershipView(int K, Collection<NodeId>, Collection<Endpoint>) (:74-89).  Registers
 * n_nodes endpoints (hostname bytes as one blob + n_nodes+1 offsets, ports, NodeIds) and makes
 * members[0..n_members) the current membership; identifiersSeen = the members' NodeIds plus
 * extra_id_hi/lo[0..n_extra) (ids of departed nodes -- never pruned, :167-201).  Builds on the GPU: the
 * K ring keys of every endpoint (:579-582), K sorted rings, observer / subject tables, configuration id. */
int rapid_view_build(rapid_engine* h, const uint8_t* hostnames, const int32_t* host_off, const int32_t* ports,
                     const int64_t* id_hi, const int64_t* id_lo, int32_t n_nodes, const int32_t* members,
                     int32_t n_members, const int64_t* extra_id_hi, const int64_t* extra_id_lo, int32_t n_extra);
/* Appends n_new endpoints (hostname bytes, port, NodeId) to the registry, as NON-members, without touching the membership, the
 * identifiers seen, detector instances or loaded streams: the nodes *first_index_out .. + n_new - 1 can then be named by
 * alerts (a receiver first hears of most joiners through an UP alert, R/MembershipService.java:677-685), passed to
 * rapid_view_ring_add, and have expected observers (:292-303).  n_max of the engine bounds the registry. */
int rapid_view_register_endpoints(rapid_engine* h, const uint8_t* hostnames, const int32_t* host_off, const int32_t* ports,
                                  const int64_t* id_hi, const int64_t* id_lo, int32_t n_new, int32_t* first_index_out);
/* isSafeToJoin(Endpoint, NodeId) (:100-115) and ringAdd(Endpoint, NodeId) (:123-160): the NodeId is explicit, as in
 * the Java; a successful ringAdd records it as node's identifier.  Order of checks as in the reference:
 * identifier already seen -> RAPID_EUUID_SEEN, else already a member -> RAPID_ENODE_EXISTS. */
int rapid_view_is_safe_to_join(rapid_engine* h, int32_t node, int64_t id_hi, int64_t id_lo, int32_t* status_out);
int rapid_view_ring_add(rapid_engine* h, int32_t node, int64_t id_hi, int64_t id_lo);
int rapid_view_ring_delete(rapid_engine* h, int32_t node);                                 /* :167-201 */
int rapid_view_observers(rapid_engine* h, int32_t node, int32_t* out, int32_t cap, int32_t* n_out);  /* :210-224 */
int rapid_view_subjects(rapid_engine* h, int32_t node, int32_t* out, int32_t cap, int32_t* n_out);   /* :267-282 */
int rapid_view_expected_observers(rapid_engine* h, int32_t node, int32_t* out, int32_t cap,
                                  int32_t* n_out);                                         /* :292-303 */
int rapid_view_ring_numbers(rapid_engine* h, int32_t observer, int32_t subject, int32_t* out, int32_t cap,
                            int32_t* n_out);                                               /* :397-418 */
int rapid_view_ring(rapid_engine* h, int32_t k, int32_t* out, int32_t cap, int32_t* n_out); /* :380-388 */
int rapid_view_ring_key(rapid_engine* h, int32_t k, int32_t node, int64_t* key_out);       /* :579-582 */
int rapid_view_is_host_present(rapid_engine* h, int32_t node, int32_t* present_out);       /* :330-337 */
int rapid_view_size(rapid_engine* h, int32_t* n_out);                                      /* :425-432 */
int rapid_view_config_id(rapid_engine* h, int64_t* id_out);                                /* :360-372, 544-556 */
/* whole tables, row-major [n_nodes][K] (observers of members = ring successors; rows of non-members hold
 * their expected observers; subjects rows of non-members are -1) -- what the scenario generators consume */
int rapid_view_tables(rapid_engine* h, int32_t* observers, int32_t* subjects, uint8_t* member, int32_t n_nodes);
/* Quirk Q4 of the reference, reproduced (R/MembershipView.java:143-152, 181-195, 210-224): getObserversOf is memoised per node,
 * and ringAdd / ringDelete drop only the entries of the changed node's ring predecessors -- an entry survives a view change that
 * moves its node's observers when the ring minimum changes (the maximum's successor wraps to it).  The only reader on this path
 * is invalidateFailingEdges (R/MultiNodeCutDetector.java:147-149), which asks for the observers of the MEMBERS in preProposal
 * -- members the round's alerts name on >= L rings ("hot").  The engine keeps the same memo on the device: a hot member's
 * observers are memoised from today's table the first round it is hot since its entry was last dropped, the round index reads
 * hot members' observers from the memo, and rapid_apply_cut / rapid_view_ring_add / _delete drop exactly the entries the Java
 * drops.  A stale entry therefore stays stale as long as the reference's does, and the tally applies the implicit reports the
 * reference would apply (tests provoke the quirk and compare with the oracle's faithful cache).  One memo per engine: the
 * population shares the view object, as the oracle's does; a real deployment has one cache per node, filled when THAT node first
 * had the subject in its preProposal -- a hot subject that no receiver ever had in preProposal at a batch end is memoised here
 * and not there.
 * rapid_view_q4_emulation(h, 0) switches the memo off: the index always reads today's observers (the round-3 behaviour).
 * rapid_view_q4_at_risk reports: for hot[0..n) (non-members and duplicates skipped) a member without an entry gets one (what its
 * first getObserversOf does), a member whose memoised observers differ from today's is written to out (capacity cap, *n_out =
 * how many there are; 0 = the quirk is not live for any of them).  rapid_sim_index_info's info[7] bit 1 says the same about the
 * round index that was last built. */
int rapid_view_q4_at_risk(rapid_engine* h, const int32_t* hot, int32_t n, int32_t* out, int32_t cap, int32_t* n_out);
int rapid_view_q4_emulation(rapid_engine* h, int32_t on);

/* ---- MultiNodeCutDetector, one instance (R/MultiNodeCutDetector.java) ---------------------------------
 * State lives on the GPU; every call runs the exact sequential kernel on one wavefront.
 * rapid_cd_aggregate <- aggregateForProposal(AlertMessage) (:76-82) applied to n alerts in order;
 * out_idx receives the concatenated returned proposals, out_counts[i] the size of alert i's return value. */
int rapid_cd_create(rapid_engine* h, int32_t K, int32_t H, int32_t L, rapid_cd** out);      /* :51-60  */
void rapid_cd_destroy(rapid_cd* cd);
int rapid_cd_aggregate(rapid_cd* cd, const rapid_alert_record* alerts, int32_t n, int32_t* out_idx, int32_t cap,
                       int32_t* out_counts, int32_t* n_out);                               /* :76-128 */
int rapid_cd_invalidate(rapid_cd* cd, int32_t* out_idx, int32_t cap, int32_t* n_out);       /* :137-164 */
int rapid_cd_num_proposals(rapid_cd* cd, int32_t* n_out);                                  /* :62-66  */
int rapid_cd_clear(rapid_cd* cd);                                                          /* :169-178 */

/* ---- whole population: MembershipService.handleMessage(BatchedAlertMessage) at every receiver ---------
 * (R/MembershipService.java:300-354 with the filter of :644-675).  records = the receivers' delivered
 * streams back to back, rec_off[r]..rec_off[r+1] (in records, rec_off[0] = 0) = receiver r; every receiver starts the round
 * with an empty detector and announcedProposal == false in the engine's current configuration.
 * The records stay what they are: 20-byte rapid_alert_records in device memory, read ONCE per round -- by the alert-tally
 * kernel itself, which compares every record's configuration id with the view's current one (another id: dropped, :653-657),
 * maps its subject to a slot of the round's index (tables in LDS; through L2 for populations whose tables do not fit) and
 * tallies it, all on the record's way through the registers.  No other pass touches a delivered record, nothing about a
 * stream is kept between rounds, and a view change needs no second look at the loaded streams.  src is never read, as in
 * R/MultiNodeCutDetector.java:101.  A stream holds at most (2^32 - 2^20) / 20 records (RAPID_ECAPACITY).
 * rapid_sim_load_streams: host records, copied into the engine's own device buffer (the PCIe transfer). */
int rapid_sim_load_streams(rapid_engine* h, const rapid_alert_record* records, const int64_t* rec_off,
                           int32_t n_receivers);
/* same, records already on this device (`records_bytes` readable bytes covering them; borrowed for the call: copied device to
 * device into the engine's buffer); d_rec_off is a device pointer to n_receivers+1 int64 and IS borrowed until the next load */
int rapid_sim_load_streams_device(rapid_engine* h, const void* d_records, uint64_t records_bytes,
                                  const int64_t* d_rec_off, int32_t n_receivers);
/* same without any copy: the engine reads the caller's device buffers IN PLACE (both borrowed until the next load / attach /
 * generate or the engine's destruction; 4-byte aligned; never written).  What a host that receives a round's deliveries into
 * device memory calls once per round: attaching costs nothing, the round is index + tally + vote count. */
int rapid_sim_attach_streams_device(rapid_engine* h, const void* d_records, uint64_t records_bytes,
                                    const int64_t* d_rec_off, int32_t n_receivers);
/* The delivered streams made ON THE DEVICE instead of loaded (rapid_sim_generate): alerts = the round's distinct
 * alerts in batch order, batch b = alerts[batch_off[b] .. batch_off[b + 1]) = one BatchedAlertMessage (never empty:
 * RAPID_EINVAL); every receiver gets every batch exactly once (the fan-out of R/UnicastToAllBroadcaster.java:46-63), receiver
 * r in an order of its own: position j holds batch perm(j), a seeded permutation of [0, n_batches) evaluated in place, in two
 * levels -- the batch list in lines of eight consecutive batches; the lines in a pseudo-random order (a two-round Feistel network
 * on a mixed-radix domain that covers the line count with less than its square root to spare, keyed by mix64(seed +
 * receivers[r]), cycle-walked into range), the eight batches of a line under an affine map of their places drawn from the line and
 * the receiver (csrc/index_kernels.h: gen_perm_at; rapid_amd/scenarios.py: hashed_order / deliver_hashed are the same statement
 * on the host).  No keys, no sort, no bound on
 * receivers x batches.  `alerts` IS the declared alert set of the round (rapid_sim_set_alert_set is implied and refused).
 * batch_keep (optional, [n_batches]): batch b reaches receiver r only if a per-(r, b) 32-bit draw is <= batch_keep[b]
 * (0xFFFFFFFF: every receiver) -- late deliveries of an earlier configuration (alerts of the set that carry another
 * configuration id; dropped per delivery, R/MembershipService.java:653-657) and lossy links; the places of a batch that does
 * not reach a receiver hold empty records (no ring number, no batch end), so every stream has batch_off[n_batches] records.
 * format RAPID_GEN_RESOLVED: 8-byte records {the subject's entry of the round index, ring mask + status + batch end} -- the
 * 20-byte records of the deliveries never exist, the tally looks nothing up; valid for the view they were generated in
 * (after a view change: RAPID_ESTATE, generate again).  RAPID_GEN_BOUNDARY: the 20-byte records themselves, exactly what a
 * load would have been handed (the synthetic input of a round, made where it is consumed).
 * rapid_sim_pass_times out[2] = the device time of the last generation. */
#define RAPID_GEN_RESOLVED 0
#define RAPID_GEN_BOUNDARY 1
int rapid_sim_generate(rapid_engine* h, const rapid_alert_record* alerts, const int64_t* batch_off, int32_t n_batches,
                       const uint32_t* batch_keep, const int32_t* receivers, int32_t n_receivers, uint64_t seed, int32_t format);
/* Optional, after a load: declares the round's DISTINCT alerts (every delivered record is a byte-identical copy of
 * one of them, flags aside -- one AlertMessage is broadcast to all receivers, R/UnicastToAllBroadcaster.java:46-52).
 * The per-round index (which subjects can reach the L watermark at all, their adjacency) and the one-time validation
 * of the alerts against the current view are then computed from these n_alerts records instead of from a pass over
 * every delivered record.  Without it the library scans the delivered streams.
 *
 * The index alone changes nothing about the semantics: every delivered record still goes through the whole filter of
 * R/MembershipService.java:644-675 (configuration id, UP / DOWN against the membership, range, rings), and a delivered
 * report about a subject / ring that the declared set does not contain -- which the index was not built for -- makes the
 * next call that reads results (rapid_sim_results, rapid_sim_count_votes, rapid_sim_round, rapid_sim_proposal) return
 * RAPID_EINVAL: the round's results are void, the set was wrong. */
int rapid_sim_set_alert_set(rapid_engine* h, const rapid_alert_record* alerts, int64_t n_alerts);
/* same, the distinct alerts already in device memory (`alerts_bytes` readable bytes covering them: n_alerts * 20 > alerts_bytes is
 * RAPID_EINVAL; 4-byte aligned; borrowed until the next load / attach / generate or declaration; never written): read in place by
 * the round index -- what a host that receives a round's deliveries into device
 * memory (rapid_sim_attach_streams_device) declares them with: no copy, no synchronisation on the round's path */
int rapid_sim_set_alert_set_device(rapid_engine* h, const void* d_alerts, uint64_t alerts_bytes, int64_t n_alerts);
/* Opt-in, per loaded stream set (a load resets it): asks the tally to treat a delivered record that fails the filter of
 * R/MembershipService.java:644-675 as an ERROR of the stream (RAPID_EINVAL, results void) instead of dropping it per delivery
 * -- the instantiation without the per-delivery filter, a few instructions per record cheaper (both read the same 8 B per
 * record).  The request is honoured on VERIFIED facts only -- nothing rests on the caller's word: (1) the load pass found the
 * view's current configuration id on every delivered record and every subject in the registry's range; (2) the marks are
 * current for the view the tally runs in; (3) every declared alert passes the filter under the current view.  If any of
 * these fails -- e.g. late deliveries of an earlier configuration are among the streams, BASELINE configs[4] -- the tally
 * silently runs the per-delivery filter instead and drops those records as the reference does (:653-657).  Checked per
 * delivery either way: UP / DOWN against the membership, rings covered by the index -> RAPID_EINVAL as above.  A delivered
 * record that passes these checks passes the reference's filter, so the results are the reference's for ANY stream that is
 * accepted, not only for byte copies of the declared alerts.
 * on == 2 adds the ONE thing on this path that rests on the caller's word: no delivered record carries another configuration id
 * (no late deliveries among them).  If, in addition, every declared alert is of the engine's configuration, the tally then does
 * not read the records' configuration ids at all -- one 8-byte load per record ({dst, word}) instead of two, the ids stay in the
 * cache lines (~3 % of the kernel at N = 10^4) -- and a late delivery that is there all the same is tallied as if it were current.
 * The library takes that path by itself, on its own knowledge, for the boundary records it lays down (rapid_sim_generate,
 * rapid_sim_round_tiled with RAPID_GEN_BOUNDARY).  Anything but 0, 1, 2: RAPID_EINVAL. */
int rapid_sim_trust_alert_copies(rapid_engine* h, int32_t on);
/* Starts another round over the streams (and the declared alert set) that are loaded: the per-round index is built again
 * by the next tally, as it is after a load.  What a round costs = index + tally + vote count; bench.py times exactly that. */
int rapid_sim_new_round(rapid_engine* h);
/* alert-tally kernel over all loaded receivers (asynchronous on the engine's stream) */
int rapid_sim_tally(rapid_engine* h);
/* per-receiver results of the last tally: index of the batch whose processing announced a proposal (-1 if
 * none), getNumProposals(), proposal size, 64-bit proposal fingerprint (0 if none).  Any pointer may be NULL. */
int rapid_sim_results(rapid_engine* h, int32_t* emit_batch, int32_t* num_proposals, int32_t* prop_count,
                      uint64_t* fingerprint, int32_t n_receivers);
/* receiver r's proposal as handed to FastPaxos.propose: sorted by the ring-0 comparator (:346-348) */
int rapid_sim_proposal(rapid_engine* h, int32_t receiver, int32_t* out, int32_t cap, int32_t* n_out);

/* ---- fast round over the population (R/FastPaxos.java:125-156) -----------------------------------------
 * Every receiver that announced a proposal votes for it; identical proposals are counted and verified element by
 * element (nothing is decided on fingerprints alone).  With a communicator every rank settles its own voters (candidate =
 * its lowest voter's proposal, verified votes, voters), ONE RCCL all-gather moves the ranks' answer blocks and every rank
 * merges them identically; only a round whose ranks hold different candidates without a quorum falls back to the
 * all-reduce of the positional vote histogram.  The quorum test N - floor((N-1)/4) uses N = current membership size. */
typedef struct rapid_round_result {
    int32_t decided;            /* 1 if some proposal reached the fast quorum */
    int32_t cut_size;           /* size of the decided proposal */
    int32_t quorum;             /* N - floor((N-1)/4) */
    int32_t membership_size;    /* N */
    int64_t votes_total;        /* receivers (all ranks) that proposed */
    int64_t votes_winner;       /* votes for the most popular proposal */
    int32_t distinct_local;     /* distinct proposals among this rank's receivers (when a quorum was confirmed without
                                 * counting the other proposals: 1 = every voter voted for the winner, 2 = at least two) */
    int32_t reserved;
    int64_t config_id;          /* configuration the round ran in */
} rapid_round_result;
int rapid_sim_count_votes(rapid_engine* h, rapid_round_result* out);
int rapid_sim_decided_cut(rapid_engine* h, int32_t* out, int32_t cap, int32_t* n_out); /* ring-0 order */
/* tally + vote count (+ apply, if decided and apply != 0): one full round; new_config_id may be NULL */
int rapid_sim_round(rapid_engine* h, int32_t apply, rapid_round_result* out, int64_t* new_config_id);
/* The same for a round whose deliveries and distinct alerts already lie in device memory, in ONE call: rapid_sim_attach_streams_device
 * + rapid_sim_set_alert_set_device (skipped if d_alerts is NULL and n_alerts 0: nothing declared) + rapid_sim_trust_alert_copies(trust)
 * + rapid_sim_round -- same checks, same errors, same state afterwards; one crossing of the host's foreign-function boundary per
 * round instead of four (the per-round sequence of R/MembershipService.java:300-354 followed by R/FastPaxos.java:94-156 and, when
 * apply != 0 and the round decides, R/MembershipService.java:385-430). */
int rapid_sim_round_device(rapid_engine* h, const void* d_records, uint64_t records_bytes, const int64_t* d_rec_off, int32_t n_receivers,
                           const void* d_alerts, uint64_t alerts_bytes, int64_t n_alerts, int32_t trust, int32_t apply,
                           rapid_round_result* out, int64_t* new_config_id);

/* ---- one round over a population that does not fit one launch (BASELINE configs[4]: 10^6 receivers x ~1.5 x 10^5 deliveries) ------
 * The receivers are taken TILE BY TILE: a tile's deliveries are made on the device from the round's alert set exactly as
 * rapid_sim_generate makes them (same seeded per-receiver orders, same batch_keep draws: receiver i's stream does not depend on the
 * tiling), the tile is tallied, and the fast-round votes are ACCUMULATED ACROSS THE TILES' LAUNCHES: the first voter's proposal is
 * the round's candidate, every later voter that holds its fingerprint is compared with it bit for bit (as a bitmap over the round's
 * hot slots; nothing is counted on fingerprints alone) and counted.  What R/FastPaxos.java:141-150 keeps per proposal is a count:
 * a tile's delivered records and node lists are gone with the next tile; every receiver's announce batch, getNumProposals(),
 * proposal size and fingerprint stay (rapid_sim_results with n_receivers = the whole population; rapid_sim_proposal for the
 * receivers of the last tile).  The round index is built once; no stream synchronisation between tiles.  With a communicator
 * every rank takes its own receivers this way and ONE all-gather at the end merges the ranks' accumulated answers (as
 * rapid_sim_count_votes does for populations held in one launch).  If the candidate ends without a quorum although the voters are
 * not unanimous, the exact plurality is owed: the receivers' fingerprints (summed over the ranks as a positional histogram) bound
 * what any proposal can have; below the quorum nothing is decided, otherwise the tiles are taken once more with the proposal of
 * the winning fingerprint as the candidate (the deliveries are a function of the seed: the same streams again).
 * tile_receivers: receivers per launch (0 = sixteen waves' worth per CU); the tile's stream buffer -- tile_receivers x
 * batch_off[n_batches] records of 8 (RAPID_GEN_RESOLVED) or 20 bytes (RAPID_GEN_BOUNDARY) -- is the only place a delivered record
 * of the round ever exists.  The library makes the copies itself and vouches for them (rapid_sim_trust_alert_copies is implied).
 * rapid_sim_round_tiled_info: out = {wall time of the last tiled round in ms, tiles launched, passes over the population (1, or 2
 * after an exact-plurality pass), records delivered / 10^6}. */
int rapid_sim_round_tiled(rapid_engine* h, const rapid_alert_record* alerts, const int64_t* batch_off, int32_t n_batches,
                          const uint32_t* batch_keep, const int32_t* receivers, int32_t n_receivers, int32_t tile_receivers,
                          uint64_t seed, int32_t format, rapid_round_result* out);
int rapid_sim_round_tiled_info(rapid_engine* h, double out[4]);

/* ---- decideViewChange (R/MembershipService.java:385-430) -----------------------------------------------
 * members in `cut` are removed (ringDelete), non-members are added (ringAdd with their registered NodeId);
 * rings, tables and the configuration id are rebuilt on the GPU; detector state of the population is cleared. */
int rapid_apply_cut(rapid_engine* h, const int32_t* cut, int32_t n, int64_t* new_config_id);

/* ---- FastPaxos fast round, one instance (R/FastPaxos.java:62-68, 125-156): host-side control object ---- */
int rapid_fast_round_create(int64_t config_id, int32_t membership_size, rapid_fast_round** out);
void rapid_fast_round_destroy(rapid_fast_round* f);
/* one FastRoundPhase2bMessage; *decided_out = 1 once a decision exists */
int rapid_fast_round_vote(rapid_fast_round* f, int32_t sender, int64_t config_id, const int32_t* endpoints,
                          int32_t n, int32_t* decided_out);
int rapid_fast_round_decision(rapid_fast_round* f, int32_t* out, int32_t cap, int32_t* n_out);

/* ---- wire ingest (host only, no device needed): serialized protobuf messages of rapid.proto -> packed records, so
 * that a live MembershipService can pass on the bytes it received (R/MembershipService.java:174-196) from a direct
 * buffer.  Endpoints inside the messages are resolved through an endpoint map built from the same registry that
 * rapid_view_build receives (hostname bytes + port -> node index). ---- */
typedef struct rapid_endpoint_map rapid_endpoint_map;
int rapid_endpoint_map_create(const uint8_t* hostnames, const int32_t* host_off, const int32_t* ports, int32_t n,
                              rapid_endpoint_map** out);
void rapid_endpoint_map_destroy(rapid_endpoint_map* m);
/* node index of (hostname, port); RAPID_ENODE_MISSING if it is not registered */
int rapid_endpoint_map_lookup(const rapid_endpoint_map* m, const uint8_t* hostname, int32_t hostname_len, int32_t port,
                              int32_t* node_out);
#define RAPID_MSG_OTHER 0
#define RAPID_MSG_BATCHED_ALERT 3 /* RapidRequest.batchedAlertMessage     (rapid.proto:26) */
#define RAPID_MSG_FAST_ROUND_2B 5 /* RapidRequest.fastRoundPhase2bMessage (rapid.proto:28) */
/* which `content` case a serialized RapidRequest carries (its field number; RAPID_MSG_OTHER for an empty request) and
 * where its payload lies inside msg */
int rapid_decode_request(const uint8_t* msg, int64_t len, int32_t* kind_out, int64_t* payload_off, int64_t* payload_len);
/* one remoting.BatchedAlertMessage (rapid.proto:95-99) -> one packed record per AlertMessage (:101-110), in message
 * order, the last one flagged RAPID_ALERT_LAST_IN_BATCH.  id_hi / id_lo (optional, parallel to out) receive the NodeId
 * of every alert (the joiner's id in UP alerts, extractJoinerUuidAndMetadata R/MembershipService.java:677-685; zeros
 * when absent).  RAPID_EINVAL: malformed bytes, ring number >= K; RAPID_ENODE_MISSING: an endpoint that is not in
 * the map; RAPID_ECAPACITY: more than cap alerts (n_out = the number needed). */
int rapid_decode_batched_alerts(const rapid_endpoint_map* m, const uint8_t* msg, int64_t len, int32_t K,
                                rapid_alert_record* out, int64_t* id_hi, int64_t* id_lo, int32_t cap, int32_t* n_out,
                                int32_t* sender_out);
/* one remoting.FastRoundPhase2bMessage (rapid.proto:124-129) -> sender, configuration id, the proposed endpoints in
 * message order (the vote that rapid_fast_round_vote takes) */
/* The join path of a live service: most receivers first hear of a joiner through an UP alert (its endpoint is in no map yet,
 * R/MembershipService.java:677-685).  rapid_decode_batched_alerts_ex decodes EVERY alert of the message and reports them
 * one by one: status[i] = RAPID_OK (out[i] is valid), RAPID_ENODE_MISSING (an endpoint of alert i is not in the map:
 * unresolved[2 i] / [2 i + 1] = offset and length, inside msg, of that serialized Endpoint; id_hi / id_lo[i] still carry the
 * alert's NodeId) or RAPID_EINVAL (malformed alert).  The call itself only fails on a malformed envelope or cap < alerts.
 * The facade registers the endpoint -- rapid_endpoint_map_add_wire here, rapid_view_register_endpoints on the engine,
 * which hands out the same index -- and decodes the message again; nothing else of the batch is lost. */
int rapid_decode_batched_alerts_ex(const rapid_endpoint_map* m, const uint8_t* msg, int64_t len, int32_t K,
                                   rapid_alert_record* out, int64_t* id_hi, int64_t* id_lo, int32_t* status, int64_t* unresolved,
                                   int32_t cap, int32_t* n_out, int32_t* sender_out);
/* appends an endpoint (next free index) unless it is there already; node_out = its index either way */
int rapid_endpoint_map_add(rapid_endpoint_map* m, const uint8_t* hostname, int32_t hostname_len, int32_t port, int32_t* node_out);
/* the same from a serialized remoting.Endpoint (what `unresolved` points at) */
int rapid_endpoint_map_add_wire(rapid_endpoint_map* m, const uint8_t* endpoint_msg, int64_t len, int32_t* node_out);
int32_t rapid_endpoint_map_size(const rapid_endpoint_map* m);
int rapid_endpoint_map_get(const rapid_endpoint_map* m, int32_t node, uint8_t* hostname_out, int32_t cap, int32_t* hostname_len,
                           int32_t* port_out);
int rapid_decode_fast_round_vote(const rapid_endpoint_map* m, const uint8_t* msg, int64_t len, int32_t* sender_out,
                                 int64_t* config_id_out, int32_t* endpoints_out, int32_t cap, int32_t* n_out);

/* ---- consensus at one node: fast round + classic-Paxos recovery (R/FastPaxos.java, R/Paxos.java) ------------
 * Host-side state machine, no device needed.  It replaces the FastPaxos object a MembershipService
 * creates per configuration (R/MembershipService.java:156-158, :427-429).  The object does no I/O: handlers append what
 * the Java would have broadcast / sent to an outbox, in the same order; the caller drains it with rapid_consensus_poll
 * and owns the network and the recovery timer (R/FastPaxos.java:107-109).
 *
 * A message is a fixed head plus its endpoint list (the value); `kind` is the RapidRequest.content field number
 * (rapid.proto:28-32).  rank_index stands in for `myAddr.hashCode()` (R/Paxos.java:102), which is not reproducible
 * across JVMs; any value that is distinct per node keeps the protocol's ordering of concurrent coordinators. */
typedef struct rapid_consensus rapid_consensus;
typedef struct rapid_rank { /* remoting.Rank, rapid.proto:133-137; ordered by round, then node_index (R/Paxos.java:333-339) */
    int32_t round;
    int32_t node_index;
} rapid_rank;
#define RAPID_MSG_PHASE1A 6 /* RapidRequest.phase1aMessage (rapid.proto:29) */
#define RAPID_MSG_PHASE1B 7 /* RapidRequest.phase1bMessage (rapid.proto:30) */
#define RAPID_MSG_PHASE2A 8 /* RapidRequest.phase2aMessage (rapid.proto:31) */
#define RAPID_MSG_PHASE2B 9 /* RapidRequest.phase2bMessage (rapid.proto:32) */
#define RAPID_DEST_BROADCAST (-1)
typedef struct rapid_consensus_msg {
    int32_t kind;        /* RAPID_MSG_FAST_ROUND_2B, RAPID_MSG_PHASE1A .. RAPID_MSG_PHASE2B */
    int32_t sender;      /* node index */
    int64_t config_id;
    rapid_rank rnd;      /* Phase1a.rank; Phase1b / Phase2a / Phase2b.rnd; zero for a fast-round vote */
    rapid_rank vrnd;     /* Phase1b.vrnd; zero otherwise */
    int32_t n_endpoints; /* length of the accompanying list: FastRoundPhase2b / Phase2b.endpoints, Phase1b / Phase2a.vval */
    int32_t dest;        /* outgoing: the node a Phase1b is sent to, RAPID_DEST_BROADCAST otherwise; ignored on input */
} rapid_consensus_msg;
/* membership_size = N of the configuration (R/FastPaxos.java:62-87); RAPID_EINVAL if it is < 1 */
int rapid_consensus_create(int32_t my_index, int32_t rank_index, int64_t config_id, int32_t membership_size,
                           rapid_consensus** out);
void rapid_consensus_destroy(rapid_consensus* c);
/* FastPaxos.propose (:95-110): registers the node's own fast-round vote with its acceptor (R/Paxos.java:246-259) and
 * queues the FastRoundPhase2bMessage broadcast.  Scheduling startClassicPaxosRound after the recovery delay is the caller's. */
int rapid_consensus_propose(rapid_consensus* c, const int32_t* proposal, int32_t n);
/* FastPaxos.handleMessages (:163-185) for one received message; a message of another configuration is ignored
 * (RAPID_OK), a kind outside 5..9 is RAPID_EINVAL (IllegalArgumentException, :181) */
int rapid_consensus_handle(rapid_consensus* c, const rapid_consensus_msg* msg, const int32_t* endpoints);
int rapid_consensus_start_classic_round(rapid_consensus* c);    /* FastPaxos.startClassicPaxosRound, :190-196 */
int rapid_consensus_start_phase1a(rapid_consensus* c, int32_t round); /* Paxos.startPhase1a, R/Paxos.java:98-111 */
/* next queued outgoing message, oldest first: *got = 0 if there is none.  RAPID_ECAPACITY (message kept, n_endpoints
 * set) if its endpoint list is longer than cap. */
int rapid_consensus_poll(rapid_consensus* c, rapid_consensus_msg* msg_out, int32_t* endpoints_out, int32_t cap,
                         int32_t* got);
/* the decided value (what onDecide receives, R/FastPaxos.java:78-85), in the order it was proposed; RAPID_ESTATE until
 * there is one.  The first decision stands (the Java asserts that there is only one). */
int rapid_consensus_decision(rapid_consensus* c, int32_t* out, int32_t cap, int32_t* n_out);
/* FastPaxos.getRandomDelayMs (:201-204): base + (long)(-1000 ln(1 - u) * N) for u = nextDouble() in [0, 1) */
int rapid_consensus_fallback_delay_ms(int32_t membership_size, int64_t base_delay_ms, double u, int64_t* delay_out);
/* Paxos.selectProposalUsingCoordinatorRule (R/Paxos.java:271-328) as a function: n_msgs Phase1b messages in arrival
 * order, message i carrying vrnd[i] and the value vvals[vval_off[i] .. vval_off[i+1]).  *chosen_out = index of a
 * message whose value is the one to propose, or -1 if no message carries a value.  RAPID_EINVAL for n_msgs < 1 (:274). */
int rapid_paxos_select_proposal(int32_t membership_size, const rapid_rank* vrnd, const int32_t* vval_off,
                                const int32_t* vvals, int32_t n_msgs, int32_t* chosen_out);
/* One classic round (round 2, one coordinator, no message loss) for a whole population whose fast round did not reach
 * its quorum -- what rapid_sim_count_votes reports as decided = 0.  Acceptor i (one per live receiver) voted in the fast
 * round iff voted[i] != 0, for the proposal identified by vote_key[i] (rapid_sim_results' 64-bit fingerprint: equal
 * proposals have equal keys; the converse is only CHECKED element-wise by rapid_sim_count_votes for the proposal it
 * reports as the winner, for the others it rests on the fingerprint); arrival[j] = the acceptor whose Phase1b reaches the
 * coordinator j-th (NULL = index order).  Equivalent to running n_acceptors rapid_consensus objects message by message
 * (tests/test_consensus.py does). */
typedef struct rapid_classic_round_result {
    int32_t decided;         /* 1 iff more than N/2 acceptors are live and one of them voted */
    int32_t chosen_acceptor; /* an acceptor whose fast-round vote is the decided value; -1 if none */
    int32_t promises_used;   /* Phase1b messages received when the coordinator chose */
    int32_t rule;            /* clause that chose: 1 one distinct value, 2 a value seen more than N/4 times, 3 any value */
    int64_t messages;        /* messages of the round: Phase1a + Phase1b + Phase2a + Phase2b */
} rapid_classic_round_result;
int rapid_classic_round_population(int32_t membership_size, int32_t n_acceptors, const uint64_t* vote_key,
                                   const uint8_t* voted, const int32_t* arrival, rapid_classic_round_result* out);
/* The same recovery with CONCURRENT coordinators and message loss (several startPhase1a in flight, R/Paxos.java:98-111; promises
 * that belong to another rank ignored, :156-188; acceptors that have promised a higher rank refusing a Phase2a, :195-216): every live
 * acceptor is a whole consensus instance (the code behind rapid_consensus_*) holding its fast-round vote, behind the reference's
 * test network -- one FIFO per destination, a broadcast queued at every live node, the sender's included (PaxosTests.java:403-476,
 * R/UnicastToAllBroadcaster.java:46-53).  starts[j] = {step, acceptor, round}: before delivery step `step` (ascending) that acceptor calls
 * startPhase1a(round); rank_index[a] stands in for myAddr.hashCode() (NULL: a + 2).  Delivery: step t hands node schedule[t] the
 * oldest message queued for it, or LOSES it if drop[t] != 0 (schedule == NULL: until nothing is queued -- or n_steps steps, if
 * n_steps > 0 -- a seeded choice among the nodes with queued messages, each message lost with probability `loss`).  Values are
 * identified by their 64-bit keys as in rapid_classic_round_population.  decided_vote_of[a] (optional, [n_acceptors]) = an acceptor
 * whose fast-round vote is what node a decided, -1 if it has not.  Memory is one queue entry per (message, destination): at most
 * RAPID_CLASSIC_ROUNDS_MAX_ACCEPTORS acceptors; the closed form above serves whole populations. */
#define RAPID_CLASSIC_ROUNDS_MAX_ACCEPTORS 4096
typedef struct rapid_classic_start {
    int32_t step, acceptor, round;
} rapid_classic_start;
typedef struct rapid_classic_rounds_result {
    int32_t decided_nodes;   /* nodes that have decided */
    int32_t agreed;          /* 1 iff every node that decided decided the same value (the protocol's safety: reported, not assumed) */
    int32_t chosen_acceptor; /* an acceptor whose fast-round vote is the decided value; -1 if nobody decided */
    int32_t lost;            /* messages lost */
    int64_t steps;           /* delivery steps taken */
    int64_t undelivered;     /* messages still queued at the end */
    int64_t sent[4];         /* Phase1a, Phase1b, Phase2a, Phase2b messages queued (a broadcast counts once per live destination) */
    int64_t delivered[4];    /* ... and handled */
} rapid_classic_rounds_result;
int rapid_classic_rounds_population(int32_t membership_size, int32_t n_acceptors, const uint64_t* vote_key, const uint8_t* voted,
                                    const int32_t* rank_index, const rapid_classic_start* starts, int32_t n_starts,
                                    const int32_t* schedule, const uint8_t* drop, int64_t n_steps, uint64_t seed, double loss,
                                    rapid_classic_rounds_result* out, int32_t* decided_vote_of);
/* wire forms of the five consensus messages (rapid.proto:124-169).  decode: the payload of a RapidRequest whose kind
 * rapid_decode_request reported; out->dest is set to RAPID_DEST_BROADCAST.  encode: a complete serialized RapidRequest,
 * byte for byte what protobuf-java emits for the messages R/Paxos.java and R/FastPaxos.java build (fields in number
 * order, zero scalars omitted, the Rank / Endpoint sub-messages always present). */
int rapid_decode_consensus_message(const rapid_endpoint_map* m, int32_t kind, const uint8_t* msg, int64_t len,
                                   rapid_consensus_msg* out, int32_t* endpoints_out, int32_t cap);
int rapid_encode_consensus_request(const rapid_endpoint_map* m, const rapid_consensus_msg* msg, const int32_t* endpoints,
                                   uint8_t* out, int64_t cap, int64_t* len_out);

/* ---- multi-GPU: one engine per rank, receivers sharded; per round one RCCL all-gather of the ranks' vote answers
 * over xGMI + the same merge on every rank (fallback: all-reduce of the vote histogram) ------- */
#define RAPID_UNIQUE_ID_BYTES 128
int rapid_comm_unique_id(uint8_t out[RAPID_UNIQUE_ID_BYTES]); /* rank 0 creates, host layer broadcasts */
int rapid_engine_comm_init(rapid_engine* h, const uint8_t id[RAPID_UNIQUE_ID_BYTES], int32_t rank, int32_t n_ranks);
/* rank and size as the communicator reports them (ncclCommUserRank / ncclCommCount); {0, 1} without a communicator */
int rapid_engine_comm_info(rapid_engine* h, int32_t* rank, int32_t* n_ranks);

/* ---- instrumentation -------------------------------------------------------------------------------------
 * stream: the hipStream_t all engine work is enqueued on (for HIP-event timing on the right stream).
 * stats[0..7] of the last tally: exact-replayed sub-chunks, fast sub-chunks, full-sweep invalidations,
 * receiver restarts, implicit reports applied, records consumed, 0, 0. */
void* rapid_engine_stream(rapid_engine* h);
int rapid_engine_sync(rapid_engine* h);
/* counters of the tally launches since the streams were loaded: {exact sub-chunks, lean windows, full invalidation
 * sweeps, restarts, implicit reports applied, records consumed, lean-path give-ups, careful sub-chunks} */
int rapid_sim_stats(rapid_engine* h, uint64_t stats[8]);
/* average duration (ms) of the tally kernel over `reps` back-to-back launches, HIP events on the engine stream */
int rapid_sim_time_tally(rapid_engine* h, int32_t reps, float* ms_avg);
/* the per-round index of the loaded streams (built on demand): info = {hot subjects, adjacency entries, waves per
 * workgroup, workgroups, LDS bytes per workgroup, alerts pre-validated (0/1), where the tally maps a record's subject to its
 * slot (boundary records: 1 = direct tables in LDS, 2 = compressed tables in LDS, 0 = tables in memory, read through L2 --
 * chosen by what fits the LDS next to the receivers' detector state; 3 = nowhere: generated resolved records carry their
 * subjects' entries), bit 0: alert set declared, bit 1: a hot member's memoised observers are stale in this round (Q4 is
 * live), bit 2: the records are pre-validated boundary records known to be of the engine's own configuration (laid down by the
 * library, or vouched for at level 2 of rapid_sim_trust_alert_copies) -- the tally leaves their configuration ids in the cache
 * lines (one load per record)}; index_ms = device time of the last index build that was timed: a call that asks (index_ms != NULL) has the
 * NEXT build bracketed by timing events -- this call's own if the index is stale; 0 before the
 * first timed build.  Rounds nobody asks about carry no timing events. */
int rapid_sim_index_info(rapid_engine* h, int32_t info[8], float* index_ms);
/* device times (ms): out[0] = the last index build, out[1] = 0 (there is no resolve pass: a delivered record is read once, by
 * the tally), out[2] = the last rapid_sim_generate, out[3] = 0 */
int rapid_sim_pass_times(rapid_engine* h, float out[4]);
/* testing / measurement knob, a bit set (0 = normal): 1 = every window through the exact sequential path, 8 = careful
 * path only (no cold / fast windows), 64 = never the pre-validated instantiation (per-delivery filter), 128 = never direct
 * tables, 256 = tables in memory, 512 = sharded vote count always through the histogram all-reduces (never the all-gather +
 * merge of the ranks' local counts), 1024 = every receiver dealt to the workgroups statically (no common pool for the last
 * eighth), 2048 = the vote count never uses the statistics the tally kernel gathers (always a counting pass), 4096 = the
 * round index built by several workgroups (the form of populations >= 40,000 nodes) whatever the size, 16384 = a view change
 * sorts all K rings again instead of compacting / merging the old ones, 131072 = the round index of a declared alert set built
 * by the two-kernel form (touch pass + one workgroup) instead of the one-launch form, 262144 = the vote verification compares the
 * voters' node lists instead of their slot bitmaps, 8192 = the packed detector state (two slots per LDS word: the form of rounds with
 * more than 4,096 hot subjects) whatever the size, 1048576 = packed rounds over boundary records keep their dictionary in LDS as hashed
 * buckets of one-byte remainders (exact; slots renumbered in hash order) instead of looking subjects up in memory, where the round
 * is eligible (every named subject hot, at most 2^21 nodes), 2097152 = a tiled round (rapid_sim_round_tiled) makes and tallies its
 * tiles strictly one after the other on one stream (by default the next tile's deliveries are made on a second stream while this
 * tile is tallied), 4194304 = boundary records known to be of the engine's own configuration still have their configuration
 * ids loaded and compared per delivery (see rapid_sim_trust_alert_copies, level 2), 8388608 = the fast round is settled INSIDE
 * the tally launch (the first workgroup to finish publishes its first voter's proposal as the candidate, every workgroup compares its
 * voters' bitmaps with it at its end, the last one lays down the answer block: no counting or verifying launch follows; measured
 * at N = 10^4: the round 1 % shorter, the tally kernel 2-3 % longer -- opt-in),
 * 32 = measurement only: stream the records through
 * the registers without tallying them (results are meaningless).  Every bit selects another PRODUCT path or instantiation
 * (all of them parity-tested); none adds code that the default does not ship. */
int rapid_sim_set_force_exact(rapid_engine* h, int32_t on);

/* ---- TEST BUILD ONLY (librapid_mi355x_test.so = the same sources compiled with -DRAPID_TEST_BUILD): testing aids and
 * measurement probes.  The product library librapid_mi355x.so does not export them, contains no probe kernel and reads no
 * environment variable. ---- */
#ifdef RAPID_TEST_BUILD
/* testing aid: n delivered records starting at `first` -- boundary records: subject and core word (ring mask, bit 14 DOWN,
 * bit 15 UP, bit 16 end of batch); resolved records: the subject's dictionary entry and the core word */
int rapid_debug_read_records(rapid_engine* h, int64_t first, int32_t n, uint32_t* subjects, uint32_t* core_words);
/* measurement probe (not a product path): stream the loaded records with the tally kernel's access pattern and no
 * processing.  Register loads: variant 0 = 2 KiB tiles x 8 in flight, 1: 4 KiB x 4, 2: 8 KiB x 2, 3: 2 KiB x 4,
 * 4: 1 KiB x 8, 5: 1 KiB x 16, 6: 1 KiB x 4; LDS-DMA loads (the tally kernel's path): 7: 1 KiB x 4, 8: 1 KiB x 8,
 * 9: 1 KiB x 6.  `waves` = waves per workgroup (16 waves per CU unless RAPID_PROBE_WAVES_PER_CU says otherwise) */
int rapid_debug_stream_probe(rapid_engine* h, int32_t variant, int32_t waves, int32_t reps, float* ms_avg);
/* measurement aid: the tally kernel's counters per workgroup ([rows][8], the rows rapid_sim_stats sums) */
int rapid_debug_block_stats(rapid_engine* h, uint64_t* out, int32_t cap_rows, int32_t* rows_out);
/* testing aids for the sharded vote count (one GPU standing in for n ranks): the answer block this engine's voters
 * contribute to rapid_sim_count_votes' all-gather (out == NULL: only *seg_bytes), and the device-side merge of n_ranks such
 * blocks laid end to end, as every rank runs it after the all-gather.  *status = 1: merged, *out filled like
 * rapid_sim_count_votes fills it; 2: the voters disagree somewhere and the general (histogram) count would run. */
int rapid_debug_vote_segment(rapid_engine* h, void* out, int64_t cap_bytes, int64_t* seg_bytes);
int rapid_debug_vote_merge(rapid_engine* h, const void* segments, int32_t n_ranks, rapid_round_result* out, int32_t* status);
/* testing aid for the test suite's own fault tolerance (tests/conftest.py): launches a kernel that stores through a wild pointer --
 * a REAL device memory fault: the runtime aborts the process.  Only called by tests/test_gpu_fault_isolation.py, and only when
 * RAPID_TEST_REAL_FAULT=1 is set by whoever runs it. */
int rapid_debug_device_fault(rapid_engine* h);

#endif /* RAPID_TEST_BUILD */

#ifdef __cplusplus
}
#endif
#endif /* RAPID_MI355X_H */
