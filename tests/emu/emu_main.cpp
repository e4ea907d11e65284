// TEST INFRASTRUCTURE ONLY: runs rapid_amd/csrc/tally_kernel.h under the SIMT emulator (tests/emu/hip/).
#include <hip/hip_runtime.h>  // resolves to tests/emu/hip/hip_runtime.h

#include "tally_kernel.h"

namespace emu {
Block* g_block = nullptr;
Dim g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;

static void trampoline() {
    Block* w = g_block;
    const int t = w->cur;
    w->body();
    w->done[t] = 1;
    w->op[t] = OP_NONE;
    swapcontext(&w->ctx[t], &w->sched);
}

// One workgroup of nthreads (= waves x 64).  Fibers run one at a time in a random order until each is parked at a
// collective; a wave-level collective resolves when all live lanes of THAT wave are parked at it, __syncthreads
// when every live fiber of the block is.  Waves therefore progress independently, as on the device.
void run_block(unsigned block, unsigned grid, unsigned nthreads, const std::function<void()>& body, uint64_t seed) {
    Block w;
    g_block = &w;
    w.body = body;
    w.nthreads = (int)nthreads;
    w.rng = seed * 0x9E3779B97F4A7C15ull + 12345;
    g_blockIdx.x = block;
    g_gridDim.x = grid;
    g_blockDim.x = nthreads;
    const int T = (int)nthreads;
    w.ctx.resize(T);
    w.stacks.resize(T);
    w.done.assign(T, 0);
    w.op.assign(T, OP_NONE);
    w.arg.assign(T, 0);
    w.aux.assign(T, 0);
    w.res.assign(T, 0);
    for (int t = 0; t < T; ++t) {
        w.stacks[t].resize(192 * 1024);
        getcontext(&w.ctx[t]);
        w.ctx[t].uc_stack.ss_sp = w.stacks[t].data();
        w.ctx[t].uc_stack.ss_size = w.stacks[t].size();
        w.ctx[t].uc_link = &w.sched;
        makecontext(&w.ctx[t], trampoline, 0);
    }
    std::vector<int> order(T), runnable(T, 1);
    for (int t = 0; t < T; ++t) order[t] = t;
    for (;;) {
        for (int i = T - 1; i > 0; --i) {
            w.rng = w.rng * 6364136223846793005ull + 1442695040888963407ull;
            std::swap(order[i], order[(w.rng >> 33) % (uint64_t)(i + 1)]);
        }
        bool ran = false;
        for (int i = 0; i < T; ++i) {
            const int t = order[i];
            if (w.done[t] || !runnable[t]) continue;
            ran = true;
            w.cur = t;
            g_threadIdx.x = (unsigned)t;
            swapcontext(&w.sched, &w.ctx[t]);
            runnable[t] = 0;  // parked (or finished)
        }
        bool all_done = true;
        for (int t = 0; t < T; ++t) all_done = all_done && w.done[t];
        if (all_done) break;
        // resolve wave-level collectives
        bool progressed = false;
        for (int wv = 0; wv < T / W; ++wv) {
            int op = OP_NONE, live = 0, parked = 0;
            bool same = true;
            uint64_t ballot = 0;
            for (int l = 0; l < W; ++l) {
                const int t = wv * W + l;
                if (w.done[t]) continue;
                ++live;
                if (runnable[t]) continue;
                ++parked;
                if (op == OP_NONE) op = w.op[t];
                if (w.op[t] != op) same = false;
                if (w.op[t] == OP_BALLOT && w.arg[t]) ballot |= 1ull << l;
            }
            if (live == 0 || parked < live) continue;
            if (!same) {
                std::fprintf(stderr, "emu: divergent wave op in wave %d\n", wv);
                std::abort();
            }
            if (op == OP_BLOCK_SYNC) continue;  // handled below
            int first = -1;
            for (int l = 0; l < W && first < 0; ++l)
                if (!w.done[wv * W + l]) first = wv * W + l;
            for (int l = 0; l < W; ++l) {
                const int t = wv * W + l;
                if (w.done[t]) continue;
                if (op == OP_BALLOT) w.res[t] = ballot;
                else if (op == OP_SHFL) { const int src = wv * W + (int)w.aux[t]; w.res[t] = w.done[src] ? 0 : w.arg[src]; }
                else if (op == OP_FIRST) w.res[t] = w.arg[first];
                else w.res[t] = 0;
                runnable[t] = 1;
            }
            progressed = true;
        }
        // block-level barrier: every live fiber parked at it
        bool all_at_sync = true, any_live = false;
        for (int t = 0; t < T; ++t) {
            if (w.done[t]) continue;
            any_live = true;
            if (runnable[t] || w.op[t] != OP_BLOCK_SYNC) all_at_sync = false;
        }
        if (any_live && all_at_sync) {
            for (int t = 0; t < T; ++t)
                if (!w.done[t]) { w.res[t] = 0; runnable[t] = 1; }
            progressed = true;
        }
        if (!progressed && !ran) {
            std::fprintf(stderr, "emu: deadlock (a wave waits at __syncthreads while another finished a different path?)\n");
            std::abort();
        }
    }
    g_block = nullptr;
}
}  // namespace emu

// dynamic LDS segment of the (single) resident block (`extern __shared__ smem[]` inside namespace rapid)
namespace rapid {
__attribute__((aligned(16))) unsigned char smem[160 * 1024];
}
using rapid::smem;

extern "C" {
int emu_window_records() { return rapid::kWin; }
int emu_tally_wave_bytes(int n_slots, int packed) { return rapid::tally_wave_bytes(n_slots, packed != 0); }
int emu_tally_shared_bytes(int mode, int n_nodes, int n_touched, int n_hot, int n_adj) { return rapid::tally_shared_bytes(mode, n_nodes, n_touched, n_hot, n_adj); }

// Runs the population kernel: `grid` persistent workgroups of `waves` waves, one workgroup at a time.
int emu_tally_run(const unsigned char* records, unsigned long long records_bytes, const long long* rec_off,
                  int n_receivers, int n_nodes, int K, int H, int L, long long cfg_id, const unsigned short* dict, const unsigned short* decl,
                  const int* node_of_slot, const unsigned short* smask,
                  const unsigned int* pairs, int n_hot, int n_adj, int* emit_batch, int* num_proposals,
                  int* prop_count, unsigned long long* fingerprint, int* props, int prop_cap, unsigned long long* stats,
                  int flags, int waves, int grid, int tables_in_lds, unsigned long long seed, const unsigned int* tbits,
                  const unsigned short* trank, const unsigned int* tent, int n_touched, unsigned long long* vote_res_out, int fmt, int packed,
                  const unsigned short* hoff, const unsigned char* hrem, const unsigned int* hmem, unsigned int hmul) {
    // packed: two slots per LDS word (PackedSlotDetector); only with the dictionary in memory (0), hashed in LDS (4) or none (3)
    if (packed && tables_in_lds != 0 && tables_in_lds != 3 && tables_in_lds != 4) return -9;
    if (tables_in_lds == 4 && (!packed || fmt != 1 || hoff == nullptr)) return -9;  // (the hashed dictionary: boundary records, packed state)
    // fmt: 0 = resident records (split / resolved below), 1 = the 20-byte boundary records themselves (kFmtBoundary)
    // tables_in_lds: 0 = dictionary in memory, 1 = direct tables in LDS, 2 = compressed tables in LDS, 3 = no dictionary: the
    // records carry their subjects' entries (kDictResolved, what resolve_records_kernel leaves in the first dword)
    const int lds = rapid::tally_shared_bytes(tables_in_lds, n_nodes, n_touched, n_hot, n_adj, packed != 0) + waves * rapid::tally_wave_bytes(n_hot, packed != 0) +
                    rapid::kBlockStatsBytes;
    if (lds > (int)sizeof(smem)) return -5;
    rapid::TallyParams p;
    // the 20-byte records as the boundary hands them over -> the split resident form (what engine.hip's
    // split_records_kernel produces); exactly sized, so that a read past a stream's end is caught by the bounds model
    const long long n_rec_all = std::min<long long>(std::max<long long>(rec_off[n_receivers], 0), (long long)(records_bytes / 20ull));  // (records_bytes: what the caller really holds)
    std::vector<unsigned int> core((size_t)n_rec_all * 2 + 2), cfgs((size_t)n_rec_all * 2 + 2);
    for (long long i = 0; i < n_rec_all; ++i) {
        unsigned int w[5];
        std::memcpy(w, records + i * 20, 20);
        cfgs[2 * i] = w[0], cfgs[2 * i + 1] = w[1];
        const bool stale = w[0] != (unsigned int)(unsigned long long)cfg_id || w[1] != (unsigned int)((unsigned long long)cfg_id >> 32);
        core[2 * i] = (w[3] >= rapid::kCoreStale ? rapid::kCoreStale - 1u : w[3]) | (stale ? rapid::kCoreStale : 0u);
        core[2 * i + 1] = rapid::core_word(w[4]);
    }
    p.core = fmt == 1 ? records : reinterpret_cast<const unsigned char*>(core.data());
    p.cfg = reinterpret_cast<const unsigned char*>(cfgs.data());
    if (fmt == 1 && tables_in_lds == 3) return -6;  // a boundary record carries its subject, not an entry
    p.rec_off = rec_off;
    p.n_receivers = n_receivers;
    p.n_nodes = n_nodes;
    p.K = K;
    p.H = H;
    p.L = L;
    p.cfg_id = cfg_id;
    p.idx.dict = dict;
    p.idx.decl = decl;
    p.idx.tbits = tbits;
    p.idx.trank = trank;
    p.idx.tent = tent;
    p.idx.n_touched = n_touched;
    // the entry table of rounds whose dictionary stays in memory (index_kernels.h: dict_entries_kernel; included further down,
    // so its statement is repeated here -- exactly sized: a lookup past the poison entry is caught by the address sanitiser)
    std::vector<unsigned int> entries((size_t)n_nodes + 1);
    for (int i = 0; i <= n_nodes; ++i) {
        unsigned int e = rapid::kEntryPoison | ((unsigned int)n_hot << 17);
        if (i < n_nodes) {
            unsigned int sl = (unsigned int)dict[i] & rapid::kSlotMask;
            if (sl == rapid::kNoSlot) sl = (unsigned int)n_hot + ((unsigned int)i & (unsigned int)(rapid::kDummySlots - 1));
            e = rapid::dict_entry((unsigned int)decl[i], sl);
        }
        entries[(size_t)i] = e;
    }
    p.idx.entries = entries.data();
    p.idx.hoff = hoff;
    p.idx.hrem = hrem;
    p.idx.hmem = hmem;
    p.idx.hmul = hmul;
    p.idx.hbits = rapid::hash_key_bits(n_nodes);
    if (tables_in_lds == 3)
        for (long long i = 0; i < n_rec_all; ++i) {
            const unsigned int d = core[2 * i];
            core[2 * i] = entries[d < (unsigned int)n_nodes ? d : (unsigned int)n_nodes];
        }
    static unsigned int error_flags[2];
    error_flags[0] = error_flags[1] = 0u;
    p.error_flags = error_flags;
    p.stream_bytes = fmt == 1 ? (unsigned long long)n_rec_all * 20ull : (unsigned long long)n_rec_all * 8ull;
    p.idx.node_of_slot = node_of_slot;
    p.idx.smask = smask;
    p.idx.pairs = pairs;
    p.idx.n_hot = n_hot;
    p.idx.n_adj = n_adj;
    p.emit_batch = emit_batch;
    p.num_proposals = num_proposals;
    p.prop_count = prop_count;
    p.fingerprint = fingerprint;
    p.props = props;
    p.prop_cap = prop_cap;
    p.stats = stats;
    p.waves_per_block = waves;
    p.flags = flags;
    p.stagger = (flags & 16) != 0 ? 1 : 0;
    // emulator-only selector (bit 9): everything beyond the first deal comes from the common pool (workgroups run one
    // after the other here, so the first one takes all of it -- the claims, the hand-over and the reset are what is tested)
    static unsigned int pool[2];
    pool[0] = pool[1] = 0u;
    p.n_static = n_receivers;
    p.pool = nullptr;
    static unsigned long long vote_acc[5];
    std::vector<unsigned long long> vote_res_buf(10 + ((size_t)prop_cap + 1 + 1) / 2 + 1, 0xDEADull);
    unsigned long long* const vote_res = vote_res_buf.data();
    for (int i = 0; i < 5; ++i) vote_acc[i] = 0ull;
    p.vote_acc = vote_acc;
    p.vote_res = vote_res;
    // the fast round settled by the launch itself (TallyParams::vote_cand) whenever a proposal's bitmap fits 64 words; emulator-only
    // selector bit 16: the earlier statistics only (lowest voter + voters, completed by vote_verify_kernel)
    static unsigned long long vote_cand[4 + 64];
    static unsigned int vote_deferred[1 + 4096];
    static unsigned long long vote_pub[10 + 4096];
    static unsigned int vote_seq_word;
    for (int i = 0; i < 4 + 64; ++i) vote_cand[i] = 0ull;
    for (int i = 0; i < 1 + 4096; ++i) vote_deferred[i] = 0u;
    vote_seq_word = 0u;
    const bool settle = (flags & 65536) == 0 && (n_hot + 63) / 64 <= 64 && prop_cap <= 8000;
    p.vote_cand = settle ? vote_cand : nullptr;
    p.vote_deferred = settle ? vote_deferred : nullptr;
    // emulator-only selector bit 18: a list of eight entries (its overflow is reachable); bit 17: see the block loop
    p.vote_deferred_cap = (flags & 262144) != 0 ? 8 : 4096;
    p.vote_publish = settle ? vote_pub : nullptr;
    p.vote_seq_out = settle ? &vote_seq_word : nullptr;
    p.vote_seq = 77u;
    // the voters' proposals as bitmaps over the hot slots (TallyParams::bitmaps), poisoned: a voter's words must all be written
    const int bitmap_words = (n_hot + 63) / 64;
    std::vector<unsigned long long> bitmaps((size_t)std::max(n_receivers, 1) * (size_t)std::max(bitmap_words, 1), 0xA5A5A5A5A5A5A5A5ull);
    p.bitmaps = bitmaps.data();
    p.bitmap_words = bitmap_words;
    if ((flags & 512) != 0 && n_receivers > grid * waves) {
        p.n_static = grid * waves;
        p.pool = pool;
    }
    for (int b = 0; b < grid; ++b) {
        // Workgroups run one after the other here, so the first one with a voter always finds the candidate unclaimed and every later
        // one finds it published.  Emulator-only selector bit 17 makes the publication LATE: once somebody has claimed the candidate it
        // reads as "not yet published" for the workgroups in between -- they hand their voters to the deferred list -- and as published
        // again for the last one, which has to compare them.
        if (settle && (flags & 131072) != 0 && vote_cand[0] != 0ull) vote_cand[1] = b == grid - 1 ? 1ull : 0ull;
        std::memset(smem, 0xCD, sizeof(smem));  // poison: the kernel must initialise what it reads
        const bool trusted = (flags & 256) != 0;  // emulator-only selector of the kTrusted instantiation
        auto run = [&](auto kern) { emu::run_block((unsigned)b, (unsigned)grid, (unsigned)waves * 64u, [&] { kern(p); }, seed + (unsigned)b); };
        const bool current = (flags & 16384) != 0;  // emulator-only: ... and of kCurrent (boundary records of the engine's configuration only)
        if (current) {
            if (!trusted || fmt != 1) return -9;
            if (packed && tables_in_lds == 0) run(rapid::tally_population_kernel<rapid::kDictMemory, true, rapid::kFmtBoundary, true, true>);
            else if (!packed && tables_in_lds == 0) run(rapid::tally_population_kernel<rapid::kDictMemory, true, rapid::kFmtBoundary, false, true>);
            else if (!packed && tables_in_lds == 1) run(rapid::tally_population_kernel<rapid::kDictDirect, true, rapid::kFmtBoundary, false, true>);
            else if (!packed && tables_in_lds == 2) run(rapid::tally_population_kernel<rapid::kDictCompressed, true, rapid::kFmtBoundary, false, true>);
            else return -9;
            continue;
        }
        if (packed) {
            if (tables_in_lds == 4) {
                if (trusted) run(rapid::tally_population_kernel<rapid::kDictHashed, true, rapid::kFmtBoundary, true>);
                else run(rapid::tally_population_kernel<rapid::kDictHashed, false, rapid::kFmtBoundary, true>);
            } else if (fmt == 1) {
                if (trusted) run(rapid::tally_population_kernel<rapid::kDictMemory, true, rapid::kFmtBoundary, true>);
                else run(rapid::tally_population_kernel<rapid::kDictMemory, false, rapid::kFmtBoundary, true>);
            } else {
                if (trusted) run(rapid::tally_population_kernel<rapid::kDictResolved, true, rapid::kFmtResident, true>);
                else run(rapid::tally_population_kernel<rapid::kDictResolved, false, rapid::kFmtResident, true>);
            }
            continue;
        }
        if (fmt == 1) {
            switch (tables_in_lds * 2 + (trusted ? 1 : 0)) {
                case 0: run(rapid::tally_population_kernel<rapid::kDictMemory, false, rapid::kFmtBoundary>); break;
                case 1: run(rapid::tally_population_kernel<rapid::kDictMemory, true, rapid::kFmtBoundary>); break;
                case 2: run(rapid::tally_population_kernel<rapid::kDictDirect, false, rapid::kFmtBoundary>); break;
                case 3: run(rapid::tally_population_kernel<rapid::kDictDirect, true, rapid::kFmtBoundary>); break;
                case 4: run(rapid::tally_population_kernel<rapid::kDictCompressed, false, rapid::kFmtBoundary>); break;
                default: run(rapid::tally_population_kernel<rapid::kDictCompressed, true, rapid::kFmtBoundary>); break;
            }
            continue;
        }
        switch (tables_in_lds * 2 + (trusted ? 1 : 0)) {
            case 0: run(rapid::tally_population_kernel<rapid::kDictMemory, false>); break;
            case 1: run(rapid::tally_population_kernel<rapid::kDictMemory, true>); break;
            case 2: run(rapid::tally_population_kernel<rapid::kDictDirect, false>); break;
            case 3: run(rapid::tally_population_kernel<rapid::kDictDirect, true>); break;
            case 4: run(rapid::tally_population_kernel<rapid::kDictCompressed, false>); break;
            case 5: run(rapid::tally_population_kernel<rapid::kDictCompressed, true>); break;
            case 6: run(rapid::tally_population_kernel<rapid::kDictResolved, false>); break;
            default: run(rapid::tally_population_kernel<rapid::kDictResolved, true>); break;
        }
    }
    for (int i = 0; i < 5; ++i)
        if (vote_acc[i] != 0ull) return -8;  // the last workgroup leaves the vote accumulators zeroed, too
    if (settle) {
        for (int i = 0; i < 4; ++i)
            if (vote_cand[i] != 0ull) return -11;  // ... and the candidate's words and the deferred count
        if (vote_deferred[0] != 0u || vote_seq_word != 77u) return -11;
        // the published copy is the answer block: res[0..9] and the candidate's {size, list}
        const int* ref = reinterpret_cast<const int*>(vote_res + 10);
        const int* pref = reinterpret_cast<const int*>(vote_pub + 10);
        for (int i = 0; i < 10; ++i)
            if (vote_pub[i] != vote_res[i]) return -12;
        const int n_list = ref[0] < 0 ? 0 : ref[0];
        if (pref[0] != ref[0] || n_list > prop_cap) return -12;
        for (int i = 0; i < n_list; ++i)
            if (pref[1 + i] != ref[1 + i]) return -12;
    }
    if (vote_res_out != nullptr) {
        for (int i = 0; i < 10; ++i) vote_res_out[i] = vote_res[i];
        // (behind them: 1 = settled by the launch, then the candidate's size and node list as 32-bit words)
        vote_res_out[10] = settle ? 1ull : 0ull;
        if (settle) {
            const int* ref = reinterpret_cast<const int*>(vote_res + 10);
            const int n_list = ref[0] < 0 ? 0 : ref[0];
            int* out32 = reinterpret_cast<int*>(vote_res_out + 11);
            out32[0] = ref[0];
            for (int i = 0; i < n_list; ++i) out32[1 + i] = ref[1 + i];
        }
    }
    if (pool[0] != 0u || pool[1] != 0u) return -7;  // the last workgroup must leave the pool words zeroed for the next launch
    // every voter's bitmap names exactly its proposal: as many bits as the proposal has nodes, and (up to the list's capacity)
    // the same nodes in the same order
    for (int r = 0; r < n_receivers; ++r) {
        if (prop_count[r] == 0) continue;
        int at = 0;
        for (int i = 0; i < bitmap_words * 64; ++i) {
            if (((bitmaps[(size_t)r * bitmap_words + (i >> 6)] >> (i & 63)) & 1ull) == 0ull) continue;
            if (i >= n_hot) return -10;
            if (at < prop_cap && props[(long long)r * prop_cap + at] != node_of_slot[i]) return -10;
            ++at;
        }
        if (prop_count[r] > 0 ? at != prop_count[r] : at <= prop_cap) return -10;
    }
    return error_flags[0] != 0u ? -1 : 0;
}

int emu_cd_run(unsigned short* state, int* scal, const unsigned char* alerts, int n_alerts, int n_nodes, int K, int H,
               int L, const int* obs, int* out_idx, int out_cap, int* out_counts, int* out_n, int mode,
               unsigned long long seed) {
    rapid::CdParams p;
    p.state = state;
    p.scal = scal;
    p.alerts = alerts;
    p.n_alerts = n_alerts;
    p.n_nodes = n_nodes;
    p.K = K;
    p.H = H;
    p.L = L;
    p.obs = obs;
    p.out_idx = out_idx;
    p.out_cap = out_cap;
    p.out_counts = out_counts;
    p.out_n = out_n;
    p.mode = mode;
    emu::run_block(0, 1, 64, [&] { rapid::cd_instance_kernel(p); }, seed);
    return 0;
}
}

// ---------------------------------------------------------------------------------------------------------------
// The round-index kernels (rapid_amd/csrc/index_kernels.h) under the same emulator.  They use static __shared__
// variables (the tally kernel carves everything from the dynamic segment): one copy per workgroup = one copy here, since
// workgroups run one after the other and every kernel initialises what it reads.
// ---------------------------------------------------------------------------------------------------------------
#undef __shared__
#define __shared__ static
#include "index_kernels.h"

extern "C" int emu_index_run(const unsigned char* alerts, long long n_alerts, int n_nodes, int K, int L, long long cfg_id,
                             const unsigned char* member, const int* obs, int chunked, int direct_budget, unsigned short* dict,
                             unsigned short* decl, int* node_of_slot, unsigned short* smask, unsigned int* pairs, int adj_cap,
                             unsigned int* tbits, unsigned short* trank, unsigned int* tent, int tent_cap, int* info_out,
                             unsigned long long seed, int* q4_rows, unsigned char* q4_valid, unsigned int* entries_out) {
    std::vector<unsigned int> work((size_t)n_nodes + 8, 0u);
    unsigned int* gmask = work.data();
    int* info = reinterpret_cast<int*>(work.data() + n_nodes);
    static unsigned long long zero_words[64];
    static unsigned int zero_flags[2];
    for (auto& z : zero_words) z = ~0ull;
    zero_flags[0] = zero_flags[1] = 7u;
    int mail[16];
    for (int& m : mail) m = -1;
    if (chunked == 2) {  // the whole index in one launch (index_fused_kernel): declared alert set, tables in LDS
        if (rapid::index_fused_lds_bytes(n_nodes) > (int)sizeof(smem)) return -6;
        std::memset(smem, 0xCD, sizeof(smem));
        emu::run_block(0u, 1u, 1024u, [&] {
            rapid::index_fused_kernel(alerts, n_alerts, cfg_id, member, obs, n_nodes, K, L, dict, decl, node_of_slot, smask, pairs, adj_cap, tbits, trank,
                                      tent, tent_cap, mail, direct_budget, zero_words, 64, zero_flags, 4242, q4_rows, q4_valid, entries_out);
        }, seed + 400);
        for (int i = 0; i < 8; ++i) info_out[i] = mail[i];
        if (mail[15] != 4242) return -2;
        for (auto z : zero_words) if (z != 0ull) return -3;
        if (zero_flags[0] != 0u || zero_flags[1] != 0u) return -4;
        return 0;
    }
    const int touch_grid = (int)std::max<long long>(1, std::min<long long>(8, (n_alerts + 255) / 256));
    for (int b = 0; b < touch_grid && n_alerts > 0; ++b)
        emu::run_block((unsigned)b, (unsigned)touch_grid, 256u, [&] {
            rapid::index_touch_kernel(alerts, n_alerts, n_nodes, (1u << K) - 1u, cfg_id, member, gmask,
                                      reinterpret_cast<unsigned int*>(info + 4));
        }, seed + 100 + (unsigned)b);
    std::vector<int> blk;
    int n_chunks = 0;
    if (chunked) {
        n_chunks = (n_nodes + rapid::kIndexChunk - 1) / rapid::kIndexChunk;
        blk.assign((size_t)2 * (size_t)std::max(n_chunks, 1), -1);
        for (int b = 0; b < n_chunks; ++b)
            emu::run_block((unsigned)b, (unsigned)n_chunks, 1024u, [&] { rapid::index_count_kernel(gmask, n_nodes, L, blk.data()); }, seed + 200 + (unsigned)b);
        for (int b = 0; b < n_chunks; ++b)
            emu::run_block((unsigned)b, (unsigned)n_chunks, 1024u, [&] {
                rapid::index_assign_kernel(gmask, member, n_nodes, L, blk.data(), dict, decl, node_of_slot, tbits, trank, tent, tent_cap);
            }, seed + 300 + (unsigned)b);
    }
    std::vector<unsigned short> edges, edge_mask;
    if (chunked) {  // the hot adjacency slot by slot (index_edges_kernel), as the engine launches it: 64 workgroups of 256
        edges.assign((size_t)16384 * rapid::kIndexEdgeStride, 0xEEEE);
        edge_mask.assign(16384, 0xEEEE);
        for (int b = 0; b < 64; ++b)
            emu::run_block((unsigned)b, 64u, 256u, [&] {
                rapid::index_edges_kernel(blk.data(), n_chunks, node_of_slot, member, obs, n_nodes, K, dict, q4_rows, q4_valid, edges.data(),
                                          edge_mask.data(), info, gmask);
            }, seed + 350 + (unsigned)b);
    }
    emu::run_block(0u, 1u, 1024u, [&] {
        rapid::index_build_block_kernel(gmask, member, obs, n_nodes, K, L, dict, decl, node_of_slot, smask, pairs, adj_cap, tbits, trank, tent,
                                        tent_cap, info, mail, chunked ? -1 : direct_budget, zero_words, 64, zero_flags, 4242,
                                        chunked ? blk.data() : nullptr, n_chunks, q4_rows, q4_valid, chunked ? edges.data() : nullptr,
                                        chunked ? edge_mask.data() : nullptr);
    }, seed + 400);
    for (int i = 0; i < 8; ++i) info_out[i] = mail[i];
    if (mail[15] != 4242) return -2;                       // the sequence word behind the answer
    for (auto z : zero_words) if (z != 0ull) return -3;    // what the round's launches expect zeroed
    if (zero_flags[0] != 0u || zero_flags[1] != 0u) return -4;
    for (int n = 0; n < n_nodes + 8; ++n) if (work[(size_t)n] != 0u) return -5;  // the work area, left clean for the next round
    return 0;
}

// index_hash_kernel: the hashed dictionary of a packed round + the renumbering of everything that carries slot numbers
extern "C" unsigned int emu_hash_multiplier(int i) { return rapid::hash_multiplier(i); }
extern "C" int emu_hash_build(const int* node_of_slot, const unsigned short* smask, const unsigned int* pairs, int n_hot, int n_adj, int n_nodes,
                              const unsigned char* member, unsigned short* hoff, unsigned char* hrem, unsigned int* hmem, int* nos_new,
                              unsigned short* smask_new, unsigned int* pairs_new, unsigned short* new_of_old, unsigned int* entries,
                              unsigned short* dict, int* answer, unsigned long long seed) {
    if (rapid::index_hash_lds_bytes(n_nodes, n_hot) > (int)sizeof(smem)) return -5;
    static int mail[16];
    for (int i = 0; i < 16; ++i) mail[i] = -1;
    std::memset(smem, 0xCD, sizeof(smem));
    emu::run_block(0, 1, 1024, [&] {
        rapid::index_hash_kernel(node_of_slot, smask, pairs, n_hot, n_adj, n_nodes, member, hoff, hrem, hmem, nos_new, smask_new, pairs_new, new_of_old,
                                 entries, dict, mail, 777);
    }, seed);
    if (mail[11] != 777) return -2;
    answer[0] = mail[12] >= 1 ? 1 : 0;
    answer[1] = mail[12] - 1;
    return 0;
}

// rapid_sim_generate's kernels (index_kernels.h): the alert set resolved once, then one workgroup per receiver lays its stream
// down (the seeded permutation evaluated in place).  boundary == 0: out = [n][2] dwords {entry, core word}, entries =
// [n_nodes + 1] dict_entry per node (the poison entry last); else out = the 20-byte records.  keep: per-batch thresholds or null.
extern "C" int emu_generate(const unsigned char* alerts, const long long* boff, int n_batches, const unsigned int* keep, const int* receivers,
                            int n_receivers, unsigned long long seed, long long cfg_id, int n_nodes, const unsigned int* entries,
                            unsigned char* out, int boundary, long long* rec_off_out, unsigned long long sd) {
    const long long A = boff[n_batches];
    std::vector<uint2> res((size_t)std::max<long long>(A, 1));
    if (!boundary) {
        const unsigned g = (unsigned)std::max<long long>(1, (A + 255) / 256);
        for (unsigned b = 0; b < g; ++b)
            emu::run_block(b, g, 256u, [&] { rapid::gen_resolve_alerts_kernel(alerts, A, cfg_id, (unsigned int)n_nodes, entries, res.data()); }, sd + b);
    }
    std::vector<uint4> bat((size_t)std::max(n_batches, 1));
    {
        const unsigned g = (unsigned)std::max(1, (n_batches + 255) / 256);
        for (unsigned b = 0; b < g; ++b) emu::run_block(b, g, 256u, [&] { rapid::gen_pack_batches_kernel(boff, n_batches, boundary ? nullptr : res.data(), keep, bat.data()); }, sd + 50 + b);
    }
    const unsigned gw = (unsigned)((n_receivers + rapid::kGenWavesPerBlock - 1) / rapid::kGenWavesPerBlock);
    for (unsigned b = 0; b < gw; ++b)
        emu::run_block(b, gw, (unsigned)rapid::kGenWavesPerBlock * 64u, [&] {
            rapid::gen_streams_kernel(res.data(), alerts, bat.data(), n_batches, keep, receivers, n_receivers, A, seed, out, boundary);
        }, sd + 100 + b);
    const unsigned g = (unsigned)((n_receivers + 1 + 255) / 256);
    for (unsigned b = 0; b < g; ++b) emu::run_block(b, g, 256u, [&] { rapid::gen_offsets_kernel(rec_off_out, n_receivers, A); }, sd + 900 + b);
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// The fast-round vote kernels (rapid_amd/csrc/vote_kernels.h): counting pass, verification (after a counting pass, or
// with the candidate taken from the statistics the tally kernel gathers), and the merge of the ranks' answers.
// ---------------------------------------------------------------------------------------------------------------
#define RAPID_DYNAMIC_LDS(name) alignas(16) static unsigned char name[(1 << 16) + 4096]
#include "vote_kernels.h"

extern "C" {
// mode 0: vote_count_local_kernel + vote_verify_kernel; mode 1: res[] as the tally kernel leaves it (lowest voter, voters) +
// vote_verify_kernel with the candidate read in place; mode 2: as mode 1, the comparison on the voters' bitmaps (bits[],
// bits_words per receiver, one thread per receiver).  block[] = res[10] followed by ref[1 + prop_cap] (as published).
int emu_vote_settle(const unsigned long long* fp, const int* prop_count, const int* props, int prop_cap, int n_receivers,
                    unsigned long long salt, int mode, unsigned long long* block, unsigned long long seed,
                    const unsigned long long* bits, int bits_words) {
    const int res_words = 10;
    std::vector<unsigned long long> dev((size_t)res_words + (size_t)(prop_cap + 2) / 2 + 1, 0ull);
    unsigned long long* res = dev.data();
    int* ref = reinterpret_cast<int*>(res + res_words);
    unsigned int errs[2] = {0u, 0u};
    if (mode == 0) {
        emu::run_block(0u, 1u, 1024u, [&] { rapid::vote_count_local_kernel(fp, prop_count, props, prop_cap, n_receivers, salt, errs, res, ref); }, seed);
    } else {
        unsigned long long voters = 0, rep = 0xFFFFFFFFull;
        for (int r = n_receivers - 1; r >= 0; --r)
            if (prop_count[r] != 0) {
                ++voters;
                rep = (unsigned long long)r;
            }
        res[0] = rep;
        res[2] = voters;
    }
    const bool by_bits = mode == 2 || mode == 3, bits_wave = mode == 3;  // 3: the comparison on bitmaps with a wave per receiver
    const int grid = std::max(1, (n_receivers * (by_bits && !bits_wave ? 1 : 64) + 1023) / 1024);
    unsigned int seq_word = 0u;
    for (int b = 0; b < grid; ++b)
        emu::run_block((unsigned)b, (unsigned)grid, 1024u, [&] {
            rapid::vote_verify_kernel(fp, prop_count, props, prop_cap, n_receivers, res + 4, ref, res + 6, res, res_words,
                                      reinterpret_cast<unsigned int*>(res + 9), block, &seq_word, 77u, mode != 0 ? 1 : 0, errs,
                                      by_bits ? bits : nullptr, bits_words, bits_wave ? 1 : 0);
        }, seed + 10 + (unsigned)b);
    if (seq_word != 77u) return -2;
    if (res[9] != 0ull) return -3;  // the packed counter is left at zero
    return 0;
}

// The fast-round votes of a TILED round, accumulated across the tiles' launches (vote_acc_pick_kernel / vote_acc_count_kernel per
// tile of `tile` receivers, vote_acc_finish_kernel at the end; engine.hip: rapid_sim_round_tiled).  The per-tile statistics the tally
// kernel gathers (lowest voter, voters) are worked out here.  target: 0 = the first voter's proposal is the candidate, else the
// proposal with this fingerprint.  block[] = res[10] followed by ref[1 + prop_cap], the layout of the all-gather.
int emu_vote_acc(const unsigned long long* fp, const int* prop_count, const int* props, int prop_cap, const unsigned long long* bits, int bits_words,
                 int n_receivers, int tile, unsigned long long target, unsigned int tally_error, unsigned long long* block, unsigned long long seed) {
    std::vector<unsigned long long> acc((size_t)rapid::kVoteAccWords + (size_t)bits_words + ((size_t)prop_cap + 1) / 2 + 1, 0ull);
    unsigned long long* const acc_bits = acc.data() + rapid::kVoteAccWords;
    int* const acc_list = reinterpret_cast<int*>(acc_bits + bits_words);
    acc[6] = target;
    unsigned int errs[2] = {tally_error, 0u};
    for (int base = 0, t = 0; base < n_receivers; base += tile, ++t) {
        const int n = std::min(tile, n_receivers - base);
        unsigned long long tile_res[10] = {0xFFFFFFFFull, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int r = n - 1; r >= 0; --r)
            if (prop_count[base + r] != 0) {
                ++tile_res[2];
                tile_res[0] = (unsigned long long)r;
            }
        // (a tile's node lists and bitmaps are tile-local in the product: the pointers start at the tile)
        emu::run_block(0u, 1u, 256u, [&] {
            rapid::vote_acc_pick_kernel(tile_res, fp + base, prop_count + base, props + (size_t)base * prop_cap, prop_cap, bits + (size_t)base * bits_words,
                                        bits_words, n, errs, acc.data(), acc_bits, acc_list);
        }, seed + 100u * (unsigned)t);
        const unsigned grid = (unsigned)std::max(1, (n * 64 + 1023) / 1024);
        for (unsigned b = 0; b < grid; ++b)
            emu::run_block(b, grid, 1024u, [&] {
                rapid::vote_acc_count_kernel(fp + base, prop_count + base, bits + (size_t)base * bits_words, bits_words, n, acc.data(), acc_bits);
            }, seed + 100u * (unsigned)t + 1u + b);
    }
    int* ref = reinterpret_cast<int*>(block + 10);
    for (unsigned b = 0; b < 8u; ++b)
        emu::run_block(b, 8u, 256u, [&] { rapid::vote_acc_finish_kernel(acc.data(), acc_list, prop_cap, block, ref); }, seed + 7u + b);
    return 0;
}

int emu_vote_merge(const unsigned long long* gathered, int n_ranks, int seg_words, int prop_cap, long long quorum,
                   unsigned long long* block, unsigned long long seed) {
    unsigned int seq_word = 0u;
    emu::run_block(0u, 1u, 256u, [&] { rapid::vote_merge_kernel(gathered, n_ranks, seg_words, 10, prop_cap, quorum, block, &seq_word, 5u); }, seed);
    return seq_word == 5u ? 0 : -2;
}
}

// ---------------------------------------------------------------------------------------------------------------
// The view kernels (rapid_amd/csrc/view_kernels.h): ring keys (XXH64 on the device), ring tables, ring compaction,
// configuration id.  The K sorts between them are a library call on the device (rocPRIM segmented radix sort of the
// sign-flipped keys); here std::stable_sort plays that part.
// ---------------------------------------------------------------------------------------------------------------
#include "view_kernels.h"
#include "node_id_sort.h"

#include <algorithm>
#include <numeric>

// The two small kernels of a view change that touch per-node flags (view_kernels.h): q4_invalidate_kernel -- the memo entries that
// ringAdd / ringDelete of nodes[] drop, with the member flags of nodes that leave cleared on the way -- and member_patch_kernel.
extern "C" int emu_view_flags(const int* subj, const int* ring, int n_members, const int* nodes, int n, int n_nodes, int K, unsigned char* valid, int self,
                              unsigned char* member_clear, unsigned char* member, const int* gone, int n_gone, const int* joined, int n_joined,
                              unsigned long long seed) {
    if (n > 0) {
        const unsigned grid = (unsigned)(((long long)n * K + 255) / 256);
        for (unsigned b = 0; b < grid; ++b)
            emu::run_block(b, grid, 256u, [&] { rapid::q4_invalidate_kernel(subj, ring, n_members, nodes, n, n_nodes, K, valid, self, member_clear); }, seed + b);
    }
    const int m = n_gone > n_joined ? n_gone : n_joined;
    if (m > 0) {
        const unsigned grid = (unsigned)((m + 255) / 256);
        for (unsigned b = 0; b < grid; ++b)
            emu::run_block(b, grid, 256u, [&] { rapid::member_patch_kernel(member, gone, n_gone, joined, n_joined); }, seed + 1000 + b);
    }
    return 0;
}

extern "C" int emu_view_build(const unsigned char* blob, const int* host_off, const int* ports, int n_nodes, int K, const int* members,
                              int n_members, const long long* ids_hi_sorted, const long long* ids_lo_sorted, int n_ids,
                              const unsigned char* keep /* nullable: members kept by a removal-only change */, long long* keys_out,
                              int* ring_out, int* obs_out, int* subj_out, long long* cfg_out, int* ring2_out, int* n_members2_out,
                              unsigned long long seed, int* obs2_out, int* subj2_out) {
    const size_t KN = (size_t)K * (size_t)n_nodes, KM = (size_t)K * (size_t)std::max(n_members, 1);
    std::vector<unsigned long long> hx_host(n_nodes + 1), hx_port(n_nodes + 1), skeys(KM), sk2(KM);
    std::vector<int> vals(KM), pos(KN + 1, -1);
    auto launch = [&](long long n_threads, unsigned threads, const std::function<void()>& body, unsigned long long sd) {
        const unsigned grid = (unsigned)std::max<long long>(1, (n_threads + threads - 1) / threads);
        for (unsigned b = 0; b < grid; ++b) emu::run_block(b, grid, threads, body, sd + b);
    };
    launch((long long)KN, 256u, [&] { rapid::ring_keys_kernel(blob, host_off, ports, n_nodes, K, keys_out, hx_host.data(), hx_port.data()); }, seed);
    std::vector<unsigned char> member(n_nodes + 1, 0);
    for (int i = 0; i < n_members; ++i) member[members[i]] = 1;
    launch((long long)K * n_members, 256u, [&] { rapid::ring_gather_kernel(keys_out, members, n_members, n_nodes, K, skeys.data(), vals.data()); }, seed + 1000);
    for (int k = 0; k < K; ++k) {  // the segmented sort
        std::vector<int> order(n_members);
        std::iota(order.begin(), order.end(), 0);
        const unsigned long long* sk = skeys.data() + (size_t)k * n_members;
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return sk[a] < sk[b]; });
        for (int i = 0; i < n_members; ++i) {
            ring_out[(size_t)k * n_members + i] = vals[(size_t)k * n_members + order[i]];
            sk2[(size_t)k * n_members + i] = sk[order[i]];
        }
    }
    launch((long long)K * n_members, 256u, [&] {
        rapid::ring_tables_kernel(ring_out, sk2.data(), keys_out, member.data(), n_nodes, n_members, K, pos.data(), obs_out, subj_out, 0);
    }, seed + 2000);
    launch((long long)KN, 256u, [&] {
        rapid::ring_tables_kernel(ring_out, sk2.data(), keys_out, member.data(), n_nodes, n_members, K, pos.data(), obs_out, subj_out, 1);
    }, seed + 3000);
    {   // one workgroup, and the same sequence cut into three slices whose (v, m) pairs are folded in order
        std::vector<unsigned long long> partial(8, 0ull);
        emu::run_block(0u, 1u, 1024u, [&] {
            rapid::config_id_kernel(ids_hi_sorted, ids_lo_sorted, n_ids, ring_out, n_members, hx_host.data(), hx_port.data(), cfg_out, partial.data());
        }, seed + 4000);
        long long cfg3 = 0;
        for (unsigned b = 0; b < 3u; ++b)
            emu::run_block(b, 3u, 256u, [&] {
                rapid::config_id_kernel(ids_hi_sorted, ids_lo_sorted, n_ids, ring_out, n_members, hx_host.data(), hx_port.data(), &cfg3, partial.data());
            }, seed + 4100 + b);
        emu::run_block(0u, 1u, 512u, [&] { rapid::config_id_final_kernel(partial.data(), 3, &cfg3); }, seed + 4200);
        if (cfg3 != cfg_out[0]) return -11;
    }
    if (keep != nullptr) {
        // a view change: `keep` = the new member flags.  Old rings minus the nodes that left, merged with the joiners by
        // their ring keys (count / scatter / join over chunks of the old rings) instead of a new sort
        std::vector<int> joiners;
        std::vector<unsigned char> was(n_nodes + 1, 0);
        for (int i = 0; i < n_members; ++i) was[members[i]] = 1;
        int m_new = 0;
        for (int n = 0; n < n_nodes; ++n) {
            m_new += keep[n] ? 1 : 0;
            if (keep[n] && !was[n]) joiners.push_back(n);
        }
        *n_members2_out = m_new;
        const int J = (int)joiners.size();
        const int n_chunks = (std::max(n_members, 1) + rapid::kRingChunk - 1) / rapid::kRingChunk;
        std::vector<int> chunk_kept((size_t)K * n_chunks, -1), jvals((size_t)K * std::max(J, 1)), jnodes((size_t)K * std::max(J, 1));
        std::vector<unsigned long long> sk3((size_t)K * (size_t)std::max(m_new, 1)), jkeys((size_t)K * std::max(J, 1)), jskeys((size_t)K * std::max(J, 1));
        for (int b = 0; b < K * n_chunks; ++b)
            emu::run_block((unsigned)b, (unsigned)(K * n_chunks), (unsigned)rapid::kRingChunk, [&] {
                rapid::ring_count_kernel(ring_out, n_members, n_chunks, keep, chunk_kept.data());
            }, seed + 5000 + (unsigned)b);
        if (J > 0) {
            // the joiners in DESCENDING node order on purpose: the sort network orders by (key, node) whatever it is handed
            std::vector<int> jrev(joiners.rbegin(), joiners.rend());
            launch((long long)K * J, 256u, [&] { rapid::ring_gather_kernel(keys_out, jrev.data(), J, n_nodes, K, jkeys.data(), jvals.data()); }, seed + 5500);
            const unsigned gs = (unsigned)(K * ((J + rapid::kJoinRun - 1) / rapid::kJoinRun));
            for (unsigned b = 0; b < gs; ++b)
                emu::run_block(b, gs, (unsigned)rapid::kJoinRun / 2u, [&] { rapid::ring_sort_runs_kernel(jkeys.data(), jvals.data(), J); }, seed + 5600 + b);
            launch((long long)K * J, 256u, [&] { rapid::ring_merge_runs_kernel(jkeys.data(), jvals.data(), J, K, jskeys.data(), jnodes.data()); }, seed + 5700);
            for (int k = 0; k < K; ++k)
                for (int i = 1; i < J; ++i)
                    if (jskeys[(size_t)k * J + i - 1] > jskeys[(size_t)k * J + i]) return -13;
        }
        std::vector<int> chunk_base((size_t)K * (n_chunks + 1), -7), chunk_lb((size_t)K * (n_chunks + 1), -7);
        for (int k = 0; k < K; ++k)
            emu::run_block((unsigned)k, (unsigned)K, 1024u, [&] {
                rapid::ring_chunk_prep_kernel(ring_out, sk2.data(), n_members, n_chunks, chunk_kept.data(), jskeys.data(), jnodes.data(), J, chunk_base.data(), chunk_lb.data());
            }, seed + 5900 + (unsigned)k);
        for (int b = 0; b < K * n_chunks; ++b)
            emu::run_block((unsigned)b, (unsigned)(K * n_chunks), (unsigned)rapid::kRingChunk, [&] {
                rapid::ring_scatter_kernel(ring_out, sk2.data(), n_members, n_chunks, keep, chunk_base.data(), chunk_lb.data(), jskeys.data(), jnodes.data(), J, ring2_out,
                                           sk3.data(), m_new);
            }, seed + 6000 + (unsigned)b);
        if (J > 0)
            launch((long long)K * J * 64, 256u, [&] {
                rapid::ring_join_kernel(ring_out, sk2.data(), n_members, n_chunks, keep, chunk_base.data(), jskeys.data(), jnodes.data(), J, K, ring2_out,
                                        sk3.data(), m_new);
            }, seed + 7000);
        for (int k = 0; k < K; ++k)  // the new rings are sorted by key
            for (int i = 1; i < m_new; ++i)
                if (sk3[(size_t)k * m_new + i - 1] >= sk3[(size_t)k * m_new + i]) return -12;
        // the tables of the new view, PATCHED (ring_patch_kernel + ring_nonmember_rows_kernel) from the old view's: obs_out / subj_out
        // are updated in place; obs2_out / subj2_out (nullable) receive them
        if (obs2_out != nullptr && m_new >= 2 && n_members >= 2) {
            std::vector<int> changed;
            for (int n = 0; n < n_nodes; ++n)
                if (was[n] && !keep[n]) changed.push_back(n);
            const int n_gone = (int)changed.size();
            for (int j : joiners) changed.push_back(j);
            std::vector<int> o2(obs_out, obs_out + (size_t)n_nodes * K), s2(subj_out, subj_out + (size_t)n_nodes * K);
            if (!changed.empty())
                launch((long long)changed.size() * K, 256u, [&] {
                    rapid::ring_patch_kernel(changed.data(), n_gone, J, ring2_out, sk3.data(), m_new, keys_out, keep, n_nodes, K, o2.data(), s2.data());
                }, seed + 8000);
            std::vector<int> nonmembers((size_t)n_nodes + 1, 0);
            launch((long long)n_nodes, 1024u, [&] { rapid::nonmember_list_kernel(keep, n_nodes, nonmembers.data()); }, seed + 8500);
            for (unsigned b = 0; b < 3u; ++b)  // (a grid smaller than the work: the kernel strides)
                emu::run_block(b, 3u, 64u, [&] {
                    rapid::ring_nonmember_rows_kernel(ring2_out, sk3.data(), m_new, keys_out, nonmembers.data(), n_nodes, K, o2.data(), s2.data());
                }, seed + 9000 + b);
            std::copy(o2.begin(), o2.end(), obs2_out);
            std::copy(s2.begin(), s2.end(), subj2_out);
        }
    }
    return 0;
}

// The incremental ring merge on hand-made keys (ONE ring): old ring (ring_in, skeys_in) sorted by (key, node), member[] = who
// stays or joins, joiners (join_nodes, join_skeys) sorted the same way -> the new ring.  Lets a test force EQUAL ring keys, which
// the 64-bit hashes of real endpoints never produce.
extern "C" int emu_ring_merge(const int* ring_in, const unsigned long long* skeys_in, int m_old, const unsigned char* member, const int* join_nodes,
                              const unsigned long long* join_skeys, int n_join, int* ring_out, unsigned long long* skeys_out, int m_new,
                              unsigned long long seed) {
    const int n_chunks = (std::max(m_old, 1) + rapid::kRingChunk - 1) / rapid::kRingChunk;
    std::vector<int> chunk_kept((size_t)n_chunks, -1);
    for (int b = 0; b < n_chunks; ++b)
        emu::run_block((unsigned)b, (unsigned)n_chunks, (unsigned)rapid::kRingChunk, [&] {
            rapid::ring_count_kernel(ring_in, m_old, n_chunks, member, chunk_kept.data());
        }, seed + (unsigned)b);
    std::vector<int> chunk_base((size_t)n_chunks + 1, -7), chunk_lb((size_t)n_chunks + 1, -7);
    emu::run_block(0u, 1u, 1024u, [&] {
        rapid::ring_chunk_prep_kernel(ring_in, skeys_in, m_old, n_chunks, chunk_kept.data(), join_skeys, join_nodes, n_join, chunk_base.data(), chunk_lb.data());
    }, seed + 50);
    for (int b = 0; b < n_chunks; ++b)
        emu::run_block((unsigned)b, (unsigned)n_chunks, (unsigned)rapid::kRingChunk, [&] {
            rapid::ring_scatter_kernel(ring_in, skeys_in, m_old, n_chunks, member, chunk_base.data(), chunk_lb.data(), join_skeys, join_nodes, n_join, ring_out, skeys_out, m_new);
        }, seed + 100 + (unsigned)b);
    if (n_join > 0) {
        const unsigned grid = (unsigned)(((long long)n_join * 64 + 255) / 256);
        for (unsigned b = 0; b < grid; ++b)
            emu::run_block(b, grid, 256u, [&] {
                rapid::ring_join_kernel(ring_in, skeys_in, m_old, n_chunks, member, chunk_base.data(), join_skeys, join_nodes, n_join, 1, ring_out, skeys_out, m_new);
            }, seed + 200 + b);
    }
    return 0;
}

// A cut's joiners sorted per ring (ring_sort_runs_kernel + ring_merge_runs_kernel): [K][n] pairs in any order -> sorted by (key, node)
extern "C" int emu_join_sort(unsigned long long* keys, int* nodes, int n, int K, unsigned long long* keys_out, int* nodes_out, unsigned long long seed) {
    const unsigned gs = (unsigned)(K * ((n + rapid::kJoinRun - 1) / rapid::kJoinRun));
    for (unsigned b = 0; b < gs; ++b)
        emu::run_block(b, gs, (unsigned)rapid::kJoinRun / 2u, [&] { rapid::ring_sort_runs_kernel(keys, nodes, n); }, seed + b);
    const unsigned gm = (unsigned)(((long long)K * n + 255) / 256);
    for (unsigned b = 0; b < gm; ++b)
        emu::run_block(b, gm, 256u, [&] { rapid::ring_merge_runs_kernel(keys, nodes, n, K, keys_out, nodes_out); }, seed + 1000 + b);
    return 0;
}

// a cut's joiner NodeIds in order (host code of the view change: node_id_sort.h)
extern "C" int emu_sort_node_ids(long long* hi, long long* lo, int n) {
    std::vector<std::pair<int64_t, int64_t>> ids((size_t)n);
    for (int i = 0; i < n; ++i) ids[(size_t)i] = {hi[i], lo[i]};
    rapid::sort_node_ids(ids);
    for (int i = 0; i < n; ++i) {
        hi[i] = ids[(size_t)i].first;
        lo[i] = ids[(size_t)i].second;
    }
    return 0;
}

// "was any of these NodeIds seen before?" answered into the mailbox word (ids_contains_publish_kernel): -> the published word
extern "C" int emu_ids_contains_publish(const long long* old_hi, const long long* old_lo, int n_old, const long long* new_hi, const long long* new_lo,
                                        int n_new, unsigned int seq, unsigned int* word_out, unsigned long long seed) {
    unsigned int acc[2] = {0u, 0u};
    unsigned int mail[16] = {0};
    const unsigned grid = (unsigned)std::max(1, (n_new + 255) / 256);
    for (unsigned b = 0; b < grid; ++b)
        emu::run_block(b, grid, 256u, [&] { rapid::ids_contains_publish_kernel(old_hi, old_lo, n_old, new_hi, new_lo, n_new, acc, mail, 13, seq); }, seed + b);
    if (acc[0] != 0u || acc[1] != 0u) return -2;  // the last workgroup leaves both words zero for the next call
    *word_out = mail[13];
    return 0;
}

// identifiersSeen merged on the device (ids_merge_kernel): two sorted lists of (high, low) -> one
extern "C" int emu_ids_merge(const long long* old_hi, const long long* old_lo, int n_old, const long long* new_hi, const long long* new_lo, int n_new,
                             long long* out_hi, long long* out_lo, unsigned long long seed) {
    const long long n = (long long)n_old + n_new;
    const unsigned grid = (unsigned)std::max<long long>(1, (n + 255) / 256);
    for (unsigned b = 0; b < grid; ++b)
        emu::run_block(b, grid, 256u, [&] { rapid::ids_merge_kernel(old_hi, old_lo, n_old, new_hi, new_lo, n_new, out_hi, out_lo); }, seed + b);
    return 0;
}

